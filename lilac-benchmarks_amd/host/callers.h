/*
 * host/callers.h -- our counterparts of the reference's remaining callers and of the on-disk
 * formats either side of the SpMV boundary.
 *
 * The reference programs are not rebuilt; each function below reproduces the call pattern of one
 * of them around `spmv_harness_` / `f_spmv_harness_`, which it receives as a function pointer so
 * that the same driver runs on libhip-spmv.so, on the oracle or on the reference's native.so.
 * Plain C, no GPU code.
 */
#ifndef SPMV_CALLERS_H
#define SPMV_CALLERS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *(*d_harness_fn)(double *, double *, double *, int *, int *, int *);
typedef void *(*f_harness_fn)(float *, float *, float *, int *, int *, int *);

/* A CSR matrix exactly as the ABI wants it: 1-based rowstr[rows+1] and colidx[nnz]. */
typedef struct csr_d {
    int rows, cols;
    int64_t nnz;
    int *rowstr, *colidx;
    double *a;
} csr_d;
typedef struct csr_f {
    int rows, cols;
    int64_t nnz;
    int *rowstr, *colidx;
    float *a;
} csr_f;
void csr_d_free(csr_d *m);
void csr_f_free(csr_f *m);

/* ---- SparseBench (a8) ------------------------------------------------------------------- */
/* CRS text format of SparseBench/big_gen.py:52-57: "{:12}{:12}" n nnz; n+1 x "{:12}" row
 * pointers; then "{:12} {:20.17f}" column/value lines.  Like the Fortran reader
 * (SRC/reference/gen_crs.f:757-789) only the first nnz entry lines are consumed, in order --
 * big_gen.py writes more (quirk N1 of SURVEY.md). */
int sb_read_crs(const char *path, csr_d *out);
int sb_write_crs(const char *path, const csr_d *m, const int *extra_cols, const double *extra_vals,
                 int64_t n_extra);
/* Seeded stand-in for big_gen.py:60-83 (the original is unseeded): n = size^3 rows, per row
 * max(1, trunc(N(5,4))) distinct uniform columns plus the diagonal when missing, sorted,
 * values |N(0,2)|; row pointers count the sample only, so the matrix the benchmark sees is what
 * the Fortran reader would make of big_gen's file: the first nnz entries in stream order. */
int sb_generate(int size, uint64_t seed, csr_d *out);
/* BiCG of SRC/reference/iter.f:18-100 as the reference runs it (no preconditioner; BOTH the 'n'
 * and the 't' product are the plain harness call, random.f:32,72; b = 1, x0 = 0, main.f:343-346).
 * hist[it] = ||r|| before iteration it+1.  Returns the iteration count (negative: not converged,
 * like the Fortran), *calls = SpMV calls made, *t_mult = seconds inside them. */
int sb_bicg(const csr_d *m, d_harness_fn spmv, int maxit, double rtol, double *hist, double *x_out,
            int64_t *calls, double *t_mult);

/* ---- bfs (a9) --------------------------------------------------------------------------- */
/* Headerless "rows cols nnz / x y" pairs (bfs/library.cc:169-184), converted exactly like
 * COOMatrix::convert (library.cc:56-80): rows shifted by one, colidx = y + 1 (quirk N2). */
int bfs_read_pairs(const char *path, csr_d *out);
int bfs_from_pairs(int rows, int cols, int64_t n, const int *x, const int *y, csr_d *out);
/* bfs/bfs.cc:36-68: one SpMV per level, frontier = (M*front != 0) & (distances == 0).
 * distances[rows]; returns the number of levels (= SpMV calls). */
int bfs_run(const csr_d *m, d_harness_fn spmv, int source, double *distances);

/* ---- pagerank (a10) --------------------------------------------------------------------- */
/* Column-normalise a 0/1 adjacency given as 1-based CSR and scale by d, in place: what
 * pagerank/main.cpp:100-111 obtains from the external mm library (normalise + scale). */
void pagerank_prepare(csr_d *m, double d);
/* main.cpp:128-149: iters x { mean; y = A x; y += (1-d)*mean; x = y; error = ||x - last|| }. */
double pagerank_run(const csr_d *m, d_harness_fn spmv, double d, int iters, double *x /* in/out */);

/* ---- parboil spmv (a11) ----------------------------------------------------------------- */
/* common_src/convert-dataset/convert_dataset.c:46-158: Matrix Market coordinate -> 1-based CSR,
 * symmetric entries mirrored in place, stable sort by row only. */
int pb_coo_to_csr(const char *mtx_path, int mirrored, int binary, csr_f *out);
int pb_read_vec(const char *path, float *v, int n);                 /* src/cpu/file.c:56-62  */
int pb_write_out(const char *path, const float *v, int n);          /* src/cpu/file.c:64-78  */
int pb_read_out(const char *path, float **v, int *n);
/* tools/compare-output:11-36: |r-c| <= 1e-4*max|r| or |r-c| < 0.002*|r|; returns mismatches. */
int pb_compare(const float *ref, const float *cmp, int n);
/* src/cpu/main.c:93-95: reps x f_spmv_harness_. */
void pb_run(const csr_f *m, f_harness_fn spmv, float *x, float *y, int reps);

/* ---- PATHSAMPLE ----------------------------------------------------- */
/*
 * The committor iteration of wales/PATHSAMPLE: Pfold.f90:190-225 (PFOLD) and NGT.f:976-1003 (NGT).  Both build a CRS
 * matrix from per-minimum neighbour lists (NVAL(J2,J1), values DMATMC / PBRANCH(J2,J1), NCOL(J1) neighbours of minimum
 * J1: Pfold.f90:193-211) and then make NPFOLD identical calls
 *     CALL SPMV_HARNESS(GPFOLD, NEWPFOLD, DVEC, ROW_PTR, COL_IND, NMIN-1)
 * -- `rows` is the temporary NMIN-1 (one row fewer than ROW_PTR describes), and NEWPFOLD / DVEC sit in the ABI's `a` /
 * `iv` positions, i.e. swapped against native.c:3 (ov, a, iv, ...): the backend sees the VECTOR as the value array
 * (indexed by nonzero) and the matrix values as x (indexed by column).  PFOLD then copies GPFOLD over NEWPFOLD
 * (Pfold.f90:222), so the array in the `a` position changes before every call; NGT does not, so nothing changes
 * between its calls.  Our arrays are sized so that the swapped roles stay inside allocated memory (the Fortran reads
 * NEWPFOLD beyond NMIN; here those entries exist and are 0).
 */
typedef struct ps_net {
    int nmin, maxcol;
    int *ncol;    /* [nmin]            NCOL(J1)                  */
    int *nval;    /* [nmin * maxcol]   NVAL(J2,J1), 1-based ids  */
    double *dmat; /* [nmin * maxcol]   DMATMC / PBRANCH(J2,J1)   */
} ps_net;
/* seeded stand-in for MAKED2's output (the LJ38 database, min.data / ts.data, is not in the repository): every minimum
 * gets 1..maxcol distinct neighbours and branching probabilities that sum to < 1 */
int ps_generate(int nmin, int maxcol, uint64_t seed, ps_net *out);
void ps_free(ps_net *n);
/* variant 0 = PFOLD (NEWPFOLD = GPFOLD after every call), 1 = NGT (no copy).  gpfold[nmin] receives the final GPFOLD;
 * *t_calls the seconds spent inside the npfold calls.  Returns the number of nonzeros, < 0 on error. */
int64_t ps_committor(const ps_net *n, d_harness_fn spmv, int npfold, int variant, double *gpfold, double *t_calls);

/* ---- Matrix Market (generic) -------------------------------------------------------------- */
/* coordinate real/integer/pattern, general/symmetric -> 1-based CSR of doubles (pattern: 1.0). */
int mm_read_csr(const char *path, csr_d *out);

/* ---- seeded graphs for bfs / pagerank when no dataset is at hand --------------------------- */
int graph_rmat(int scale, int edge_factor, uint64_t seed, csr_d *out); /* values 1.0, row-sorted */
int graph_grid2d(int nx, int ny, csr_d *out);

#ifdef __cplusplus
}
#endif
#endif
