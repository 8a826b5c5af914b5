c     tests/native/f_kat.f -- a Fortran object calls the library the way the headline benchmark does
c     (/root/reference/NPB3.3.1/CG/cg.f:532  call spmv_harness(q,a,p,rowstr,colidx,rows)  on vectors that
c     live in COMMON blocks, cg.f:69-80).  Test infrastructure; own code, nothing taken from cg.f but the
c     shape of the call.
c       1. the known answer of libspmv/test.cpp:44-54 (12 x 12, 4 entries, 8 empty rows);
c       2. a product of NPB class S's size (1400 rows, row lengths 8..127) whose matrix, source vector and
c          result lie in COMMON: compared entry by entry with the sequential loop of
c          libspmv/native-impl.c:1-12 written here in Fortran, bound 2 n eps sum|a x| per row
c          (tests/rowtol.py derives the bound; the GPU kernels add in another order);
c       3. the same call 20 more times with p and q swapped roles as conj_grad does (cg.f:532,628), each
c          result scaled to max |.| = 1, so that the library meets the same three matrix pointers and both
c          static vectors again; the test compares the printed |p|^2 between libraries (1e-10 relative).
c     Prints one line per part and "f_kat: success"; any mismatch ends with STOP 1.
      program fkat
      implicit none
      integer na, nz
      parameter (na = 1400, nz = 180000)
      common / main_int_mem / colidx, rowstr
      integer colidx(nz), rowstr(na+1)
      common / main_flt_mem / a, p, q
      double precision a(nz), p(na+2), q(na+2)
      double precision qref(na), s, sabs, tol, worst
      integer rows, i, j, k, len, seed, it, nbad
      external spmv_harness

c     ---- part 1: the reference's own known answer -------------------------------------------------
      a(1) = 5.0d0
      a(2) = 8.0d0
      a(3) = 3.0d0
      a(4) = 6.0d0
      rowstr(1) = 1
      rowstr(2) = 2
      rowstr(3) = 2
      rowstr(4) = 3
      rowstr(5) = 3
      rowstr(6) = 3
      do i = 7, 12
         rowstr(i) = 4
      end do
      rowstr(13) = 5
      colidx(1) = 1
      colidx(2) = 4
      colidx(3) = 2
      colidx(4) = 12
      do i = 1, 6
         p(i) = 1.0d0
         p(i+6) = 2.0d0
      end do
      do i = 1, 12
         q(i) = -1.0d0
         qref(i) = 0.0d0
      end do
      qref(1) = 5.0d0
      qref(3) = 8.0d0
      qref(6) = 3.0d0
      qref(12) = 12.0d0
      rows = 12
      call spmv_harness(q, a, p, rowstr, colidx, rows)
      nbad = 0
      do i = 1, 12
         if (q(i) .ne. qref(i)) nbad = nbad + 1
      end do
      write (*, '(a,i3)')
     >     'f_kat: known answer 12 x 12, mismatches =', nbad
      if (nbad .ne. 0) stop 1

c     ---- part 2: a class-S-sized matrix in COMMON --------------------------------------------------
      seed = 12345
      k = 0
      rowstr(1) = 1
      do i = 1, na
         len = 8 + mod(i * 37, 120)
         do j = 1, len
            k = k + 1
            seed = mod(seed * 1103 + 12345, 1048573)
            colidx(k) = 1 + mod(seed, na)
            seed = mod(seed * 1103 + 12345, 1048573)
            a(k) = dble(seed - 524286) / 524286.0d0
         end do
         rowstr(i+1) = k + 1
      end do
      do i = 1, na + 2
         seed = mod(seed * 1103 + 12345, 1048573)
         p(i) = dble(seed) / 1048573.0d0 - 0.25d0
         q(i) = 0.0d0
      end do
      rows = na
      call spmv_harness(q, a, p, rowstr, colidx, rows)
      nbad = 0
      worst = 0.0d0
      do i = 1, na
         s = 0.0d0
         sabs = 0.0d0
         do j = rowstr(i), rowstr(i+1) - 1
            s = s + a(j) * p(colidx(j))
            sabs = sabs + abs(a(j) * p(colidx(j)))
         end do
         qref(i) = s
         tol = 2.0d0 * dble(rowstr(i+1) - rowstr(i))
     >         * 1.1102230246251565d-16 * sabs
         if (abs(q(i) - s) .gt. tol) nbad = nbad + 1
         if (sabs .gt. 0.0d0)
     >      worst = max(worst, abs(q(i) - s) / sabs)
      end do
      write (*, '(a,i7,a,i8,a,i5,a,1pe10.2)') 'f_kat: rows =', na,
     >     ' entries =', k, ' rows beyond the bound =', nbad,
     >     ' worst |dq| / sum|a p| =', worst
      if (nbad .ne. 0) stop 1

c     ---- part 3: conj_grad's call pattern: q = A p, then p = A q, same matrix pointers -------------
      do it = 1, 10
         call spmv_harness(q, a, p, rowstr, colidx, rows)
         s = 0.0d0
         do i = 1, na
            s = max(s, abs(q(i)))
         end do
         do i = 1, na
            q(i) = q(i) / s
         end do
         call spmv_harness(p, a, q, rowstr, colidx, rows)
         s = 0.0d0
         do i = 1, na
            s = max(s, abs(p(i)))
         end do
         do i = 1, na
            p(i) = p(i) / s
         end do
      end do
      s = 0.0d0
      do i = 1, na
         s = s + p(i) * p(i)
      end do
      if (.not. (s .ge. 0.0d0 .and. s .lt. 1.0d300)) stop 1
      write (*, '(a,1pe23.15)')
     >     'f_kat: 20 alternating products, |p|^2 =', s
      write (*, '(a)') 'f_kat: success'
      end
