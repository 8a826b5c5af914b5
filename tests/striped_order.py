"""The summation order of k_striped (csrc/spmv_kernels.hip), restated in numpy -- addition by addition.

STRIPED does not add a row's products in the reference's left-to-right order; it adds them in an order that the LAYOUT
fixes (csrc/spmv_striped.c) and the kernel follows with no freedom left: no atomics, no floating-point contraction (the
kernel is compiled with `#pragma clang fp contract(off)`), every tree written out.  This module walks a layout built by
the host builder exactly as a wave does:

  per 256-entry sub-window, lane l holds entries 4 l .. 4 l + 3:
    p[k]   = fl(value * x)                                   (fillers and pads multiply with the zero column)
    tailv  = sum of the lane's products after its last row end     (t = t + p[k]; reset to 0 at a row-end flag)
    (sv, sf) = segmented inclusive scan of (tailv, lane has a row end) over the wave:
               row_shr:1, 2, 4, 8 inside rows of 16 lanes, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3;
               one step: v = f ? v : pv + v;  f |= pf;  a lane without a source reads pv = 0, pf = 0
    sv     = sf ? sv : sv + carry                             (carry: the open sum of the previous sub-window)
    cin    = sv of the lane below (lane 0: carry);  carry = sv of lane 63
    run    = cin;  for k: run = run + p[k];  at a row-end flag: emit run as the (row, stripe) sum, run = 0
  per (row, stripe) sum, in stripe order of the group:   acc[row] = acc[row] + sum
  per row, in group order (k_group_sum; nothing to do with one group):   y = 0; y = y + partial[g]

so that the GPU's result can be compared with it BIT FOR BIT (tests/test_gpu_parity.py), which a tolerance cannot do: one
mis-summed segment of one ulp would show.  Test infrastructure only.
"""
import ctypes

import numpy as np

from test_plan import StripedHost, striped_build

LANES = np.arange(64)


def _scan_step(v, f, pv, pf):
    return np.where(f, v, pv + v), f | pf


def _row_shr(v, n, fill):
    out = np.full_like(v, fill)
    ok = (LANES % 16) >= n
    out[:, ok] = v[:, LANES[ok] - n]
    return out


def _segscan(v, f):
    zero = v.dtype.type(0)
    for n in (1, 2, 4, 8):
        v, f = _scan_step(v, f, _row_shr(v, n, zero), _row_shr(f, n, False))
    pv, pf = np.zeros_like(v), np.zeros_like(f)
    pv[:, 16:32], pf[:, 16:32] = v[:, 15:16], f[:, 15:16]  # row_bcast:15, row_mask 0xa
    pv[:, 48:64], pf[:, 48:64] = v[:, 47:48], f[:, 47:48]
    v, f = _scan_step(v, f, pv, pf)
    pv, pf = np.zeros_like(v), np.zeros_like(f)
    pv[:, 32:64], pf[:, 32:64] = v[:, 31:32], f[:, 31:32]  # row_bcast:31, row_mask 0xc
    return _scan_step(v, f, pv, pf)


def striped_exact(lib, rs, ci, a, x, xbytes=131072, wpb=16, wgs=256, wave_rows=1024, groups=0):
    """y = A x in k_striped's own order of additions, from the layout the host builder makes with these options (the
    library's defaults).  Returns None when the matrix is not representable."""
    rows = rs.size - 1
    dtype = a.dtype
    rc, t = striped_build(lib, rs, ci, a, xbytes, wpb, wgs, wave_rows, groups)
    if rc != 0:
        return None
    try:
        nr, GM, NG, W = t.n_ranges, t.GM, t.NG, t.W
        tptr = np.ctypeslib.as_array(t.tptr, (nr * (GM + 1),)).astype(np.int64).reshape(nr, GM + 1)
        rrow = np.ctypeslib.as_array(t.rrow, (nr * 2,)).astype(np.int64).reshape(nr, 2)
        wgd = np.ctypeslib.as_array(t.wgd, (t.n_wg * 4,)).astype(np.int64).reshape(t.n_wg, 4)
        col = np.ctypeslib.as_array(t.col, (t.n_packed + 1,))[: t.n_packed].astype(np.int64)
        ctype = ctypes.c_double if dtype == np.float64 else ctypes.c_float
        val = np.ctypeslib.as_array(ctypes.cast(t.val, ctypes.POINTER(ctype)), (t.n_packed + 1,))[: t.n_packed]
        sub = 64 * t.EPL
        G = 16 // dtype.itemsize
        e = np.arange(t.n_packed, dtype=np.int64)
        r_ = e % sub  # undo the lane interleave of the values inside each sub-window
        val = val[e // sub * sub + (r_ % t.EPL) // G * (64 * G) + (r_ // t.EPL) * G + (r_ % t.EPL) % G]
        s0 = np.repeat(wgd[:, 0], t.WPB)
        ns = np.repeat(wgd[:, 1] - wgd[:, 0], t.WPB)
        grp = np.repeat(wgd[:, 2], t.WPB)
        nrows = rrow[:, 1] - rrow[:, 0]
        acc = np.zeros((nr, int(nrows.max()) + 1), dtype)
        xpad = np.concatenate([np.asarray(x, dtype), np.zeros(W + 1, dtype)])  # lanes of pads index past x: their factor is 0
        ar = np.arange(sub, dtype=np.int64)
        zero = dtype.type(0)
        for si in range(GM):
            start, nsub = tptr[:, si], (tptr[:, si + 1] - tptr[:, si]) // sub
            live = si < ns
            rowpos = np.zeros(nr, np.int64)
            carry = np.zeros(nr, dtype)
            for j in range(int(nsub[live].max()) if live.any() else 0):
                w = np.nonzero(live & (j < nsub))[0]
                P = (start[w] + j * sub)[:, None] + ar[None, :]
                c = col[P]
                f = (c >> 15).astype(bool)
                lc = c & 0x7fff
                real = lc < W
                xv = np.where(real, xpad[np.where(real, (s0[w] + si)[:, None] * W + lc, 0)], zero)
                p = (val[P] * xv).reshape(w.size, 64, t.EPL)
                f3 = f.reshape(w.size, 64, t.EPL)
                tail = np.zeros((w.size, 64), dtype)
                for k in range(t.EPL):
                    tail = np.where(f3[:, :, k], zero, tail + p[:, :, k])
                sv, sf = _segscan(tail, f3.any(axis=2))
                sv = np.where(sf, sv, sv + carry[w][:, None])
                run = np.concatenate([carry[w][:, None], sv[:, :63]], axis=1)
                carry[w] = sv[:, 63]
                emit = np.zeros((w.size, 64, t.EPL), dtype)
                for k in range(t.EPL):
                    run = run + p[:, :, k]
                    emit[:, :, k] = run
                    run = np.where(f3[:, :, k], zero, run)
                cs = np.cumsum(f, axis=1)
                ridx = rowpos[w][:, None] + cs - 1
                wi = np.broadcast_to(w[:, None], f.shape)
                acc[wi[f], ridx[f]] = acc[wi[f], ridx[f]] + emit.reshape(w.size, sub)[f]  # one sum per (row, stripe): no duplicates
                rowpos[w] += cs[:, -1]
            assert np.array_equal(rowpos[live], nrows[live])
        partial = np.zeros((NG, rows), dtype)
        for u in range(nr):
            partial[grp[u], rrow[u, 0]:rrow[u, 1]] = acc[u, : nrows[u]]
        if NG == 1:
            return partial[0]
        y = np.zeros(rows, dtype)
        for g in range(NG):
            y = y + partial[g]
        return y
    finally:
        lib.spmv_striped_free.argtypes = [ctypes.POINTER(StripedHost)]
        lib.spmv_striped_free(ctypes.byref(t))
