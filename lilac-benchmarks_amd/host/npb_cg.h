/*
 * host/npb_cg.h -- our counterpart of the NPB CG caller.
 *
 * The reference's Fortran program is not rebuilt; this C library reproduces
 * what it does around the SpMV boundary so that the backend can be measured and
 * verified on a box that has neither the reference nor a Fortran compiler:
 *
 *   npb_makea      the class matrix, bit-identical to makea/sparse
 *                  (/root/reference/NPB3.3.1/CG/cg.f:650-905) but built by
 *                  bucketing + stable sorting instead of ordered insertion, so
 *                  classes C and D take seconds / minutes instead of hours
 *   npb_cg_run     the benchmark proper (cg.f:193-402): one untimed conj_grad,
 *                  niter timed ones, 26 spmv_harness_ calls each (cg.f:532,628),
 *                  the zeta verification (cg.f:122-163,363-368) and the MOp/s
 *                  formula (cg.f:395-402)
 *
 * The SpMV is called through a function pointer with the ABI's signature so the
 * same driver runs on libhip-spmv.so, on the oracle, or on the reference's own
 * native.so.
 */
#ifndef NPB_CG_H
#define NPB_CG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *(*spmv_harness_fn)(double *ov, double *a, double *iv, int *rowstr, int *colidx, int *rows);

typedef struct npb_class {
    char name;    /* 'S','W','A','B','C','D','E' */
    int na;
    int nonzer;
    int niter;
    double shift;
    double rcond;
    double zeta_verify;
} npb_class;

typedef struct npb_matrix {
    npb_class cls;
    int64_t nnz;
    int *rowstr;   /* [na+1], 1-based */
    int *colidx;   /* [nnz],  1-based */
    double *a;     /* [nnz]           */
    double t_gen;  /* seconds spent generating */
    int row_begin; /* rows [row_begin,row_end) of the class matrix are held (0-based);          */
    int row_end;   /* npb_makea gives [0,na); rowstr is local to the block, colidx stays global */
} npb_matrix;

#define NPB_MAX_ITER 128
typedef struct npb_result {
    char cls;
    int na, niter;
    int64_t nnz;
    int64_t spmv_calls;
    double zeta, rnorm;
    double zeta_hist[NPB_MAX_ITER];
    double rnorm_hist[NPB_MAX_ITER];
    double err;         /* |zeta - verify| / verify           */
    int verified;       /* err <= 1e-10 (cg.f:363-368)          */
    double t_init;      /* s: untimed first conj_grad           */
    double t_bench;     /* s: "Time in seconds"                 */
    double t_spmv;      /* s: inside the harness calls (timed part) */
    double mops;        /* cg.f:395-402                         */
} npb_result;

/* 0 on success, -1 for an unknown class letter */
int npb_class_lookup(char name, npb_class *out);
/* Generates the class matrix.  0 on success; -1 unknown class; -2 out of memory;
 * -3 the class does not fit int32 indices (class E). */
int npb_makea(char cls, npb_matrix *out);
/* One of `nparts` contiguous row blocks of about equal weight (the multi-GPU row partition of
 * SURVEY.md 8(e)); only that block is generated.  Same values, bit for bit, as the rows of npb_makea. */
int npb_makea_part(char cls, int part, int nparts, npb_matrix *out);
/* An explicit row range [row_begin,row_end), 0-based. */
int npb_makea_rows(char cls, int row_begin, int row_end, npb_matrix *out);
void npb_matrix_free(npb_matrix *m);
/* Runs the benchmark with `spmv` as the sparse product.  niter_override > 0 shortens the timed loop
 * (verification is then skipped). */
int npb_cg_run(const npb_matrix *m, spmv_harness_fn spmv, int niter_override, int verbose,
               npb_result *out);
/*
 * The same benchmark on a backend that offers the opt-in device-resident CG step
 * (include/spmv_hip.h, spmv_hip_cg_*): the vectors stay on the GPU, the host sees only the two
 * norms per outer iteration.  The function pointers are resolved by the caller (dlsym / ctypes) so
 * that this driver does not link against the backend.
 */
typedef struct npb_cg_device_fns {
    void *(*matrix_create)(const void *a, const int *rowstr, const int *colidx, int rows, int elem_size,
                           int device, int row_begin, int row_end, int cols_hint);
    void (*matrix_destroy)(void *m);
    void *(*cg_create)(void *m);
    void (*cg_destroy)(void *c);
    int (*cg_fill_x)(void *c, double value);
    int (*cg_conj_grad)(void *c, int cgitmax, double *rnorm);
    int (*cg_norms)(void *c, double *x_dot_z, double *z_dot_z);
    int (*cg_x_from_z)(void *c, double scale);
    /* the row-partitioned step (spmv_hip_partition_rows, spmv_hip_cg_create_multi); only npb_cg_run_device_parts uses them */
    void (*partition_rows)(const int *rowstr, int rows, int parts, int *bounds);
    void *(*cg_create_multi)(void **parts, int nparts);
} npb_cg_device_fns;
int npb_cg_run_device(const npb_matrix *m, const npb_cg_device_fns *f, int niter_override, int verbose,
                      npb_result *out);
/* The same with the matrix in `nparts` row blocks of about equal nonzeros, block k on device k % ndevices (x replicated, the
 * scalars of conj_grad exchanged through the host, the slices of p peer to peer: csrc/spmv_cg.hip). */
int npb_cg_run_device_parts(const npb_matrix *m, const npb_cg_device_fns *f, int nparts, int ndevices, int niter_override,
                            int verbose, npb_result *out);

/* Prints the result in the layout of NPB's print_results (common/print_results.f). */
void npb_print_result(const npb_result *r);

#ifdef __cplusplus
}
#endif
#endif
