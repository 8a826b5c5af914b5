/*
 * include/spmv_hip.h -- optional control / introspection / device-resident
 * API of libhip-spmv.so.  Plain C-ABI: pointers, sizes and ints only.
 *
 * None of these is needed by the reference's callers (they only bind the two
 * symbols of spmv_harness.h); they exist for measurement (bench.py, the
 * drivers under lilac-benchmarks_amd/host), for the parity tests, and for the
 * row-partitioned multi-GPU path.  Where a function takes over a job that the
 * reference GPU backend does implicitly, the reference lines are cited.
 */
#ifndef SPMV_HIP_H
#define SPMV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kernel / plan identifiers -------------------------------------------- */
enum {
    SPMV_HIP_PLAN_AUTO = 0,    /* adaptive: chosen per matrix from its row-length profile   */
    SPMV_HIP_PLAN_ORDERED = 1, /* one lane per row, sequential, no FMA: bit-identical to     */
                               /* libspmv/native-impl.c:1-25                                */
    SPMV_HIP_PLAN_STREAM = 2,  /* row-segment kernel: coalesced a/colidx -> products in LDS  */
    SPMV_HIP_PLAN_VECTOR = 3,  /* L lanes per row straight from global (L = plan_param)      */
    SPMV_HIP_PLAN_BINNED = 4,  /* rows binned by length, one VECTOR launch per bin           */
    SPMV_HIP_PLAN_TILED = 5,   /* column stripes of x held in LDS, one row per lane; sums each  */
                               /* row stripe by stripe in stored order: bit-identical to the    */
                               /* reference when a row's columns are sorted                     */
    SPMV_HIP_PLAN_STRIPED = 6, /* 2-D: a workgroup = (group of column stripes, row block); one stripe */
                               /* of x in LDS at a time, row sums carried in registers across the  */
                               /* group, one partial sum per (group, row), added in group order     */
    SPMV_HIP_PLAN_HUB = 8,     /* the CSR stream in fixed chunks, segmented wave scan, the most referenced ("hub") columns   */
                               /* of x in LDS and the next 262 144 in a dense table in memory: for power-law graphs (bfs,   */
                               /* pagerank), where a few thousand columns take most of the gathers; values are read from    */
                               /* the CSR copy itself (a value refresh costs this plan nothing)                             */
    SPMV_HIP_PLAN_BLOCKED = 7  /* two phases over (column stripe, row bin) cells: a workgroup keeps a    */
                               /* stripe of x in LDS and writes the PRODUCTS binned by row; one wave per */
                               /* bin adds them in LDS.  No gathers from memory: for short rows with     */
                               /* scattered columns over a long x (SparseBench), where the CSR kernels   */
                               /* fetch a 64-byte line per entry and STRIPED has nothing to gain         */
};

/* Opaque device-resident matrix (CSR block + execution plan) on one GPU. */
typedef struct spmv_hip_matrix spmv_hip_matrix;

typedef struct spmv_hip_info {
    int32_t rows;        /* rows held by this object (row_end - row_begin)                   */
    int32_t cols;        /* inferred max(colidx)  (libspmv/gpu.c:217-223 does this on host)  */
    int64_t nnz;         /* nonzeros held                                                    */
    int32_t elem_size;   /* 8 = double, 4 = float                                            */
    int32_t device;      /* HIP device ordinal                                               */
    int32_t plan;        /* SPMV_HIP_PLAN_* actually in use                                  */
    int32_t plan_param;  /* lanes per row for VECTOR; nnz per workgroup for STREAM           */
    int32_t min_row_len, max_row_len;
    double mean_row_len;
    int32_t n_blocks;    /* STREAM: row blocks                                               */
    int32_t n_long_rows; /* rows routed to the long-row kernels                              */
    int32_t n_launches;  /* kernel launches per SpMV                                         */
    int32_t reserved;
    int64_t algorithmic_bytes; /* nnz*(elem+4) + (rows+1)*4 + cols*elem + rows*elem */
    int64_t device_bytes;      /* HBM actually allocated for this matrix                     */
} spmv_hip_info;

/* Call-level accounting of the ABI path (spmv_harness_ / f_spmv_harness_). */
typedef struct spmv_hip_stats {
    int64_t calls;          /* ABI calls served                                             */
    int64_t uploads;        /* matrix uploads (cache misses; libspmv/gpu.c:236-262)          */
    int64_t invalidations;  /* cached copies dropped because the host arrays were written    */
    double t_init;          /* s: lazy one-time device init                                  */
    double t_upload;        /* s: matrix H2D + analysis + plan                               */
    double t_x_h2d;         /* s: per-call x upload        (libspmv/gpu.c:264)               */
    double t_kernel;        /* s: per-call kernel(s), host-observed                          */
    double t_y_d2h;         /* s: per-call y download      (libspmv/gpu.c:285)               */
    double t_total;         /* s: entry -> return, all calls                                 */
    int64_t host_calls;     /* of `calls`: products dispatched to the host loop (SPMV_HIP_DISPATCH) */
    double t_host;          /* s: spent in those                                             */
    int64_t value_refreshes; /* cached copies whose VALUES were uploaded again while the sparsity pattern stayed */
                            /* (only `a` had changed: Pfold.f90:219-223); their time is part of t_upload       */
} spmv_hip_stats;

/* ---- runtime ------------------------------------------------------------- */
int spmv_hip_device_count(void);            /* visible GPUs; 0 when there is none (no abort)  */
int64_t spmv_hip_device_free_bytes(int device); /* free device memory right now (hipMemGetInfo); -1 on error */
const char *spmv_hip_version(void);
const char *spmv_hip_last_error(void);      /* message of the last failed int-returning call  */

/* ---- device-resident matrices --------------------------------------------- */
/*
 * Uploads rows [row_begin,row_end) of a host CSR matrix (1-based rowstr and
 * colidx, exactly as handed to spmv_harness_) to `device` and plans its
 * execution.  elem_size 8 = double, 4 = float.  cols_hint <= 0 infers
 * max(colidx) over the WHOLE matrix (not just the block) so that partitions
 * agree on the length of x.  Returns NULL on failure (see spmv_hip_last_error).
 * Replaces libspmv/gpu.c:66-85 (allocation) and :236-240 (upload).
 */
spmv_hip_matrix *spmv_hip_matrix_create(const void *a, const int *rowstr, const int *colidx, int rows,
                                        int elem_size, int device, int row_begin, int row_end,
                                        int cols_hint);
void spmv_hip_matrix_destroy(spmv_hip_matrix *m);
int spmv_hip_matrix_info(const spmv_hip_matrix *m, spmv_hip_info *out);
/*
 * New values for the same sparsity pattern: copies the block's share of `a` (the array spmv_hip_matrix_create was given,
 * same base pointer, same layout) into the device copy that exists -- no allocation, no validation, the plan stays.  A
 * re-laid-out plan packs its second copy again: STRIPED / TILED in place, from `colidx` when given (the caller's array, as
 * at creation) or from the device's CSR copy when NULL; BLOCKED frees its layout and builds it anew on the device (count,
 * plan, scatter: allocations, and a device-to-host copy of one counter per cell -- about the cost of its first build);
 * HUB holds no values and does nothing.  0 on success; after a failure (-1: the error string says why) the matrix is left on
 * the CSR kernels, never on a plan without its layout.
 */
int spmv_hip_matrix_refresh_values(spmv_hip_matrix *m, const void *a, const int *colidx);
/* Force a plan (tests, benchmarks).  plan_param: lanes for VECTOR (1,2,4,..,64), 0 = default. */
int spmv_hip_matrix_set_plan(spmv_hip_matrix *m, int plan, int plan_param);

/*
 * y_dev[0..rows) = A * x_dev, both DEVICE pointers on the matrix's device,
 * enqueued on `stream` (a hipStream_t; NULL = the library's own stream for
 * that device).  Asynchronous.  Replaces libspmv/gpu.c:270-282.
 *
 * ONE product in flight per matrix handle: the STRIPED plan's partial sums and the long-row kernels'
 * partials are scratch buffers of the matrix, rewritten by every launch, so two products of the same
 * handle on different streams (or from two host threads) must be ordered by the caller (the ABI path
 * serialises its calls itself).  Different handles are independent.
 */
int spmv_hip_spmv_device(spmv_hip_matrix *m, const void *x_dev, void *y_dev, void *stream);
/*
 * Times `reps` back-to-back SpMVs with HIP events recorded on the stream the
 * kernels are launched on; *ms_total receives the elapsed milliseconds.
 */
int spmv_hip_time_device(spmv_hip_matrix *m, const void *x_dev, void *y_dev, int reps, void *stream,
                         float *ms_total);

/* One process driving several GPUs: `reps` products of each of n device-resident matrices (the row blocks of one matrix,
 * one per GPU), launched rep by rep, each on its own device's stream; ms_each[i] = elapsed on matrix i's stream. */
int spmv_hip_time_multi(spmv_hip_matrix **ms, void *const *x_dev, void *const *y_dev, int n, int reps, float *ms_each);

/*
 * The traffic of one ABI call around the kernel (libspmv/gpu.c:264,285), `reps` times on one stream between two
 * HIP events: x_host -> x_dev (cols elements), y = A x, y_dev -> y_host (the matrix's rows).  For the row-
 * partitioned path y_host is the address of this block's slice of the caller's y (the host gather).
 */
int spmv_hip_time_call(spmv_hip_matrix *m, const void *x_host, void *y_host, void *x_dev, void *y_dev, int reps,
                       void *stream, float *ms_total);
/* pinned host memory for the above (and for callers that want DMA transfers): allocate, or pin in place */
void *spmv_hip_host_alloc(size_t bytes);
void spmv_hip_host_free(void *p);
int spmv_hip_host_register(void *p, size_t bytes);
void spmv_hip_host_unregister(void *p);

/* ---- opt-in device-resident CG step (csrc/spmv_cg.hip) ---------- */
/*
 * NPB's conj_grad (NPB3.3.1/CG/cg.f:447-644) with p, q, r, z, x kept on the GPU: 25 x {SpMV,
 * p.q, z/r update + r.r, p update} without a host round trip, then r = A z and ||x - r||.
 * Additional entry points only: spmv_harness_ is unchanged and callers that ignore these lose
 * nothing.  FP64, square matrix, one GPU.
 */
typedef struct spmv_hip_cg spmv_hip_cg;
spmv_hip_cg *spmv_hip_cg_create(spmv_hip_matrix *m);
/* The same step over `nparts` row partitions of one square FP64 matrix (round 4): parts[k] = rows [b_k, b_k+1) in order, every
 * part created with the full column range (cols_hint = the order of the matrix), each on its own GPU (or wrapped onto fewer).
 * Each part keeps its row slices of x, z, q, r and a full copy of p; per CG iteration the parts exchange two scalars through
 * pinned host memory (added in part order: reproducible) and their slices of p by hipMemcpyPeerAsync -- no collective library.
 * The handle is used with every other spmv_hip_cg_* entry point as a one-GPU handle is. */
spmv_hip_cg *spmv_hip_cg_create_multi(spmv_hip_matrix **parts, int nparts);
void spmv_hip_cg_destroy(spmv_hip_cg *c);
int spmv_hip_cg_set_x(spmv_hip_cg *c, const double *x_host);   /* upload x[0..rows) */
int spmv_hip_cg_fill_x(spmv_hip_cg *c, double value);          /* x = value (cg.f:216-218) */
int spmv_hip_cg_get(spmv_hip_cg *c, int which, double *host);  /* download 0:x 1:z 2:p 3:q 4:r */
int spmv_hip_cg_conj_grad(spmv_hip_cg *c, int cgitmax, double *rnorm);
int spmv_hip_cg_norms(spmv_hip_cg *c, double *x_dot_z, double *z_dot_z); /* cg.f:323-327 */
int spmv_hip_cg_x_from_z(spmv_hip_cg *c, double scale);                  /* cg.f:343-345 */

/* ---- row partition for multi-GPU (pure host arithmetic, no GPU needed) ----- */
/*
 * Splits rows into `parts` contiguous blocks of about equal nnz.
 * bounds[0] = 0 <= bounds[1] <= ... <= bounds[parts] = rows.
 */
void spmv_hip_partition_rows(const int *rowstr, int rows, int parts, int *bounds);

/* ---- thin device-memory helpers (so ctypes/C drivers need no HIP headers) --- */
void *spmv_hip_malloc(int device, size_t bytes);
void spmv_hip_free(int device, void *p);
int spmv_hip_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes);
int spmv_hip_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes);
int spmv_hip_memset(int device, void *dst_dev, int value, size_t bytes);
int spmv_hip_synchronize(int device);

/* Geometry switches used by tools/kbench.py and tools/shapebench.py (not a stable interface; they affect matrices and
 * plans created afterwards, and must not change while an upload is in flight): 0 XCD-contiguous block order of
 * k_stream / k_striped (default 1); 1 non-temporal loads of a/colidx (0); 2-4 TILED geometry (waves per workgroup, LDS
 * bytes per stripe, workgroups aimed for); 5 register sets of k_tiled; 9, 11, 12 STRIPED geometry (workgroups aimed for,
 * stripe groups, most rows per wave range; 0 = choose); 15 STRIPED kernel shape (0 = chosen per matrix, 1 = 16 waves x
 * 128 KiB of x, 2 = 8 waves x 64 KiB); 17 who converts to STRIPED (0 = the GPU, from the CSR copy in HBM, the host threads
 * when it must; 1 = the host; 2 = the GPU or fail); 18 BLOCKED: rows per bin of layouts built from now on (0 = sized to fill whole rounds of phase 2; 1024 ... 2560 in steps of 256); 19 STRIPED: per mille of weight
 * moved from the workgroups of even XCDs to those of odd ones (default 0: measured, within noise); 20 STRIPED: the packed
 * stream is loaded non-temporally 0 = when it exceeds the Infinity Cache, 1 = always, 2 = never; 21-23 HUB layouts built from
 * now on: per cent of the entries that the LDS table's columns must take for the automatic choice to keep the plan (default 25),
 * bytes of the dense second-tier table in memory (default 2 MB, 0 = none), most columns in the LDS table (< 0: what fits).  Diagnostics (ablation bits, clock stamps, fault injection) are not in this
 * library: they live in the debug build, include/spmv_hip_debug.h. */
void spmv_hip_set_tuning(int key, int value);

/* ---- ABI-path control ------------------------------------------------------ */
void spmv_hip_get_stats(spmv_hip_stats *out);
void spmv_hip_reset_stats(void);
/* Drops every cached device copy and disarms write detection (also runs at exit / dlclose). */
void spmv_hip_release(void);
/* Info of the matrix most recently used through the ABI (0 on success, -1 if none). */
int spmv_hip_abi_matrix_info(spmv_hip_info *out);

#ifdef __cplusplus
}
#endif
#endif
