/*
 * include/spmv_harness.h -- the drop-in boundary of libhip-spmv.so.
 *
 * These two symbols are the whole interface the reference's applications
 * (NPB CG, SparseBench, bfs, pagerank, parboil spmv, PATHSAMPLE) bind to.
 * They replace, one for one:
 *
 *   spmv_harness_     /root/reference/libspmv/native.c:3-6
 *                     (GPU twin being replaced: libspmv/gpu.c:211-289)
 *   f_spmv_harness_   /root/reference/libspmv/native.c:8-11
 *                     (GPU twin being replaced: libspmv/gpu.c:291-369)
 *   inner prototypes  /root/reference/libspmv/native-impl.h:4-5
 *
 * Contract:
 *   - C linkage, default visibility, trailing underscore = Fortran mangling of
 *     `call spmv_harness(q, a, p, rowstr, colidx, rows)` (NPB3.3.1/CG/cg.f:532).
 *   - every argument is passed by reference; `rows` may point at a temporary.
 *   - CSR with 1-based `rowstr[rows+1]` and 1-based `colidx[nnz]`,
 *     nnz = rowstr[*rows] - rowstr[0]; row i covers a[rowstr[i]-1 .. rowstr[i+1]-2].
 *   - computes ov[i] = sum_j a[j] * iv[colidx[j]-1] for i in [0,*rows); every
 *     ov[i] is overwritten (empty row -> 0.0); nothing else is written.
 *   - the number of columns is not passed: it is inferred as max(colidx), and
 *     iv is read only at indices [0, max(colidx)).
 *   - all arrays are caller-owned HOST memory; the callee may keep the
 *     pointers only as cache keys.  The matrix may change between calls
 *     (other pointers, or the same pointers with edited contents).
 *   - the return value is ignored by every caller; NULL is returned.
 *   - no error channel: on an unrecoverable error (no GPU, HIP failure) a
 *     diagnostic is written to stderr and abort() is called, mirroring the
 *     assert() convention of libspmv/gpu.c:42-58.  There is NO CPU fallback.
 */
#ifndef SPMV_HARNESS_H
#define SPMV_HARNESS_H

#ifdef __cplusplus
extern "C" {
#endif

void *spmv_harness_(double *ov, double *a, double *iv, int *rowstr, int *colidx, int *rows);
void *f_spmv_harness_(float *ov, float *a, float *iv, int *rowstr, int *colidx, int *rows);

#ifdef __cplusplus
}
#endif
#endif
