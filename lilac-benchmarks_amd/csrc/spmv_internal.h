/*
 * csrc/spmv_internal.h -- declarations shared by the C host side
 * (spmv_harness.c, spmv_plan.c) and the thin HIP layer (spmv_device.hip,
 * spmv_kernels.hip).  C-compatible on purpose: the host side is plain C.
 */
#ifndef SPMV_INTERNAL_H
#define SPMV_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "spmv_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- geometry constants of the kernels (one place, used by host and device) -- */
#define SPMV_WG 256               /* threads per workgroup = 4 wavefronts of 64             */
#define SPMV_STREAM_NPT 8         /* nonzeros per thread per row block in the STREAM kernel  */
#define SPMV_STREAM_CAP (SPMV_WG * SPMV_STREAM_NPT) /* = 2048 nonzeros per row block         */
#define SPMV_STREAM_ROWCAP 512    /* rows per row block (row pointers staged in LDS)          */
#define SPMV_LONG_SEG 2048        /* nonzeros per workgroup in the long-row kernel            */
#define SPMV_PAD_ELEMS 4096       /* a/colidx are over-allocated by this many entries         */
                                  /* (a = 0, colidx = 1) so aligned vector loads may overrun  */
#define SPMV_NBINS 4              /* BINNED plan: lanes per row 1 / 4 / 16 / 64               */
#define SPMV_TILE_LDS_BYTES 32768 /* TILED plan: bytes of x per stripe (two stripes live in LDS)  */
#define SPMV_TILE_WPB 8           /* TILED plan: wavefronts per workgroup (512 threads)        */
#define SPMV_TILE_CHUNK 256       /* TILED plan: products staged per wave per pass             */
#define SPMV_TILE_PAD 4096        /* TILED plan: over-allocation of the packed arrays          */
#define SPMV_TILE_TARGET_BLOCKS 512 /* two 80-KiB workgroups per CU on 256 CUs                  */

/* ---- host-side row statistics (spmv_plan.c) -------------------------------- */
typedef struct spmv_row_stats {
    int rows;
    int64_t nnz;
    int min_len, max_len;
    int empty_rows;
    double mean_len;
    double cv_len;       /* stddev / mean of the row length                              */
    int64_t hist[33];    /* hist[0]: len 0; hist[k]: 2^(k-1) <= len < 2^k                 */
} spmv_row_stats;

/* rowstr is 1-based CSR as given by the caller; rows [r0,r1). */
void spmv_plan_row_stats(const int *rowstr, int r0, int r1, spmv_row_stats *st);

/*
 * STREAM plan.  Row blocks are runs of whole consecutive rows whose nonzeros
 * total <= cap_nnz and whose count <= cap_rows; a row longer than cap_nnz forms
 * a block of its own that the stream kernel skips and the long-row kernels take.
 *
 *   blk[2*b+0] = first row of block b (local, 0-based)
 *   blk[2*b+1] = first nonzero of block b (local, 0-based)
 *   entry nblk is the sentinel {rows, nnz}.
 *
 *   seg[4*s+0] = row, seg[4*s+1] = first nnz, seg[4*s+2] = end nnz,
 *   seg[4*s+3] = partial-sum slot, or -1 when the row has one segment and the
 *                kernel stores y directly.
 *   lrow[3*k+0] = row, lrow[3*k+1] = first slot, lrow[3*k+2] = end slot
 *                 (rows with > 1 segment only; reduced in slot order, which
 *                 keeps the result reproducible).
 */
typedef struct spmv_stream_plan {
    int nblk;
    int *blk;
    int nseg;
    int *seg;
    int nlrow;
    int *lrow;
    int nslots;
    int n_long_rows;
} spmv_stream_plan;

int spmv_plan_stream_build(const int *rowstr, int r0, int r1, int cap_nnz, int cap_rows, int long_seg,
                           spmv_stream_plan *out);
void spmv_plan_stream_free(spmv_stream_plan *p);

/*
 * BINNED plan: rows sorted (stably, so neighbours stay neighbours) into
 * SPMV_NBINS length classes; bin_rows holds the row ids bin after bin,
 * bin_off[b]..bin_off[b+1] delimits bin b.  Rows longer than cap_nnz are left
 * out (long-row kernels take them; the stream plan lists them).
 */
typedef struct spmv_bin_plan {
    int bin_off[SPMV_NBINS + 1];
    int *bin_rows;
} spmv_bin_plan;

int spmv_plan_bins_build(const int *rowstr, int r0, int r1, int cap_nnz, spmv_bin_plan *out);
void spmv_plan_bins_free(spmv_bin_plan *p);
int spmv_plan_bin_of_len(int len);    /* 0..SPMV_NBINS-1 */
int spmv_plan_bin_lanes(int bin);     /* 1, 4, 16, 64    */

/*
 * TILED plan (spmv_tiled.c): the block re-laid-out so that a stripe of x can live in LDS.
 * See the header of spmv_tiled.c for the layout.
 */
typedef struct spmv_tiled_host {
    int elem;          /* 8 / 4                                                        */
    int W;             /* columns per stripe                                            */
    int S;             /* stripes                                                       */
    int WPB;           /* waves per workgroup                                           */
    int RW;            /* rows per wave (<= 64)                                         */
    int B;             /* workgroups (row blocks)                                       */
    int n_groups;      /* B * WPB                                                       */
    int64_t n_packed;  /* entries incl. alignment pads                                  */
    int64_t nonempty_chunks; /* (group, stripe) chunks that hold at least one entry          */
    void *val;         /* T[n_packed + SPMV_TILE_PAD]                                   */
    uint16_t *col;     /* stripe-local column                                           */
    int32_t *wptr;     /* [n_groups*S + 1]                                              */
    uint16_t *seg;     /* [n_groups*S*(RW+1)]                                           */
} spmv_tiled_host;

int spmv_tiled_geometry(int rows, int cols, int elem_size, int target_blocks, spmv_tiled_host *t);
int spmv_tiled_build(const void *a, const int *rowstr, const int *colidx, int r0, int r1, int cols,
                     int elem_size, int target_blocks, spmv_tiled_host *t);
void spmv_tiled_free(spmv_tiled_host *t);
void spmv_tiled_set_wpb(int wpb);       /* experiments: 8 (default) or 16 waves per workgroup */
void spmv_tiled_set_xbytes(int bytes);  /* experiments: 32768 (default) or 65536 bytes of x per stripe */
void spmv_tiled_set_blocks(int blocks); /* experiments: workgroups aimed for (0 = default) */

/*
 * STRIPED plan (spmv_striped.c): 2-D decomposition, (stripe group, row block) per workgroup, a stripe of x in
 * LDS at a time, row ends flagged in bit 15 of the 16-bit column, row sums carried in registers across the
 * stripes of a group, one partial sum per (stripe group, row).
 */
#define SPMV_STRIPE_XBYTES 131072  /* bytes of x per stripe (single LDS buffer, one workgroup/CU) */
#define SPMV_STRIPE_WPB 16         /* waves per workgroup                                        */
#define SPMV_STRIPE_TARGET_WGS 256 /* workgroups aimed for: one 160-KiB workgroup per CU          */
#define SPMV_STRIPE2_XBYTES 65536  /* the half-width shape: 64 KiB of x, 8 waves, two workgroups per CU (2 x 80 KiB of LDS) */
#define SPMV_STRIPE2_WPB 8
#define SPMV_STRIPE_NT_BYTES (320ll << 20) /* a packed stream well beyond the 256-MB Infinity Cache is loaded non-temporally (crossover measured between 300 and 360 MB: profiles/r03_striped_nt_by_stream_size.log) */
#define SPMV_STRIPE_XCD_BIAS 0     /* per mille of weight moved from even-XCD workgroups to odd-XCD ones (tuning key 19) */
#define SPMV_STRIPE_EPL 4          /* entries per lane per sub-window                            */
#define SPMV_STRIPE_NACC 16        /* row accumulators per lane                                   */
#define SPMV_STRIPE_WAVE_ROWS (64 * SPMV_STRIPE_NACC) /* rows a wave can carry across a stripe group */
typedef struct spmv_striped_opts {
    int xbytes;     /* LDS bytes of x per stripe                                     */
    int wpb;        /* waves per workgroup                                           */
    int target_wgs; /* workgroups aimed for (NB * NG)                                */
    int wave_rows;  /* most rows per wave range (a multiple of 64)                   */
    int groups;     /* stripe groups, 0 = choose                                     */
    int xcd_bias;   /* per mille of weight moved from the workgroups of even XCDs to those of odd ones (0 = none) */
    int stripes;    /* column stripes, 0 = choose (the fewest that fit the LDS buffer, or a few more so that
                       they divide into the groups)                                  */
} spmv_striped_opts;
typedef struct spmv_striped_host {
    int elem, W, S, WPB, EPL;
    int NB, NG, GM;  /* most row blocks of a group, stripe groups, most stripes per group             */
    int n_wg;        /* workgroups = all groups' row blocks                                           */
    int n_ranges;    /* n_wg * WPB wave ranges                                                        */
    int cap;         /* elements of the kernel's LDS buffer; cap - 1 = the zero column               */
    int skewed;      /* one tile dwarfs the others (a dense row): the automatic choice declines      */
    int64_t n_packed, n_dummies, max_tile;
    void *val;
    uint16_t *col;
    int32_t *tptr; /* [n_ranges * (GM + 1)] tile starts, see spmv_striped.c        */
    int32_t *rrow; /* [n_ranges * 2] first and end row of each wave range          */
    int32_t *wgd;  /* [n_wg * 4] first stripe, end stripe, group, row block of it  */
} spmv_striped_host;

void spmv_striped_default_opts(spmv_striped_opts *o);
/* which kernel shape suits the matrix: 1 = 16 waves over 128-KiB stripes (one workgroup per CU), 2 = 8 waves over 64-KiB
 * stripes (two per CU) */
int spmv_striped_choose_shape(const spmv_row_stats *st, int elem_size, int cols);
void spmv_striped_geometry(int rows, int cols, int elem_size, const spmv_striped_opts *o, spmv_striped_host *t);
/* the pieces of the conversion that are decisions, not data movement: shared by the host builder and the device builder
 * (spmv_striped_gpu.hip), which therefore produce the same layout byte for byte */
void spmv_striped_plan_geometry(int rows, int cols, int elem_size, int64_t nnz, const spmv_striped_opts *opts, spmv_striped_host *t);
void spmv_striped_alloc_blocks(const spmv_striped_opts *opts, int rows, const int64_t *gent, int *gnb, spmv_striped_host *t);
void spmv_striped_cut_all(const spmv_striped_opts *opts, int rows, const uint32_t *wgt, const int *gnb, spmv_striped_host *t);
int spmv_striped_tile_starts(const int64_t *tsize, const int64_t *rfill, int64_t nnz, spmv_striped_host *t);
/* the conversion on the GPU, from the device-resident CSR block (spmv_striped_gpu.hip): 0 (t: geometry and host copies of
 * tptr / rrow / wgd; the packed arrays in device memory), -2 / -3 as the host builder, -4 = not a case for it (the caller
 * runs the host builder), -1 a HIP error */
int spmv_striped_build_device(const int *d_rowptr, const int *d_col, const void *d_val, int rows, int cols, int elem_size,
                              int64_t nnz, const spmv_striped_opts *opts, void *stream, spmv_striped_host *t,
                              void **d_tcol_out, void **d_tval_out);
/* opts == NULL: the defaults */
int spmv_striped_build(const void *a, const int *rowstr, const int *colidx, int r0, int r1, int cols,
                       int elem_size, const spmv_striped_opts *opts, spmv_striped_host *t);
void spmv_striped_free(spmv_striped_host *t);

/* Adaptive choice (SPMV_HIP_PLAN_AUTO): fills *plan / *param from the row statistics. */
void spmv_plan_choose(const spmv_row_stats *st, int elem_size, int cols, int *plan, int *param);

/* ---- kernel launchers (spmv_kernels.hip); stream is a hipStream_t ---------- */
typedef struct spmv_dev_csr {
    const int *rowptr;   /* [rows+1], 0-based, local to the block                         */
    const int *colidx;   /* [nnz + pad], 1-based column ids (as given by the caller)       */
    const void *val;     /* [nnz + pad]                                                   */
    int rows;
    int64_t nnz;
    int elem_size;
} spmv_dev_csr;

int spmvk_ordered(const spmv_dev_csr *m, const void *x, void *y, void *stream);
int spmvk_vector(const spmv_dev_csr *m, int lanes, const int *row_list, int nrows, const void *x,
                 void *y, void *stream);
int spmvk_stream(const spmv_dev_csr *m, const int *blk, int nblk, const void *x, void *y, void *stream);
int spmvk_long(const spmv_dev_csr *m, const int *seg, int nseg, const int *lrow, int nlrow,
               void *partials, const void *x, void *y, void *stream);
/* TILED: device copies of the spmv_tiled_host arrays; x is the caller's 0-based vector */
typedef struct spmv_dev_tiled {
    const int32_t *wptr;
    const uint16_t *seg;
    const uint16_t *col;
    const void *val;
    int elem, W, S, WPB, RW, B, rows, cols;
} spmv_dev_tiled;
int spmvk_tiled(const spmv_dev_tiled *t, const void *x, void *y, void *stream);
typedef struct spmv_dev_striped {
    const int32_t *tptr;
    const int32_t *rrow;
    const int32_t *wgd;
    const uint16_t *col;
    const void *val;
    void *partial;          /* T[NG * rows], NULL when NG == 1 */
    int elem, W, S, WPB, NB, NG, GM, EPL, n_wg, n_ranges, wave_rows, rows, cols;
    int xbytes; /* LDS stripe capacity the layout was built for */
    int64_t n_packed; /* entries of the packed stream, pads included */
} spmv_dev_striped;
/* BLOCKED (spmv_blocked.hip): two phases over (column stripe, row bin) cells.  Phase 1 streams the entries stripe-major
 * with the stripe of x in LDS and writes the products bin-major; phase 2 streams the products, one wave per bin, into the
 * bin's slice of y in LDS */
typedef struct spmv_dev_blocked {
    void *val1;        /* T[n1] phase-1 order (stripe, bin, row), lane-interleaved inside windows of 256 entries    */
    uint16_t *col1;    /* [n1] stripe-local column; pads point to the zero column W - 1                           */
    uint32_t *dest4;   /* [n1 / 4] phase-2 position / 4 of every four products                                     */
    int *wg;           /* [n_wg * 3] phase-1 workgroups: stripe, first window, end window                          */
    void *prod;        /* T[n2] the products in phase-2 order (bin, stripe, row): scratch, rewritten by every call */
    uint16_t *lrow;    /* [n2] bin-local row of every product                                                      */
    int *pw, *pb, *ps; /* phase 2's pieces: [n_pieces + 1] first window, [n_pieces] bin, [n_pieces] slot of its partial sums or -1 */
    int *jn;           /* [n_join * 3] cut bins: bin, first slot, slots                                             */
    double *part;      /* [n_slots * RB] partial sums of the pieces of cut bins: scratch                           */
    int elem, rows, cols, S, wcols, W, RB, B, n_wg, n_pieces, n_slots, n_join;
    int runs;          /* phase 2 sums runs of one row's products before the LDS add: set when a row is long enough to have them */
    int64_t nnz, n1, n2;
    size_t bytes;
} spmv_dev_blocked;
/* the host side of the layout (spmv_blocked_plan.c): geometry and, from the entries of every cell, where the cells lie */
#define SPMV_BLOCKED_WIN 256           /* entries per window: 64 lanes x 4 */
#define SPMV_BLOCKED_MAX_STRIPES 2048  /* per-wave cursors of the device builder in LDS */
#define SPMV_BLOCKED_MAX_BIN_ROWS 2560 /* two workgroups of 4 waves x RB doubles per CU */
typedef struct spmv_blocked_plan {
    int S, wcols, W, RB, B;               /* stripes, columns per stripe, LDS elements, rows per bin, bins */
    int n_wg, n_pieces, n_slots, n_join;
    int64_t n1, n2;                       /* entries of the two streams, pads included */
    uint32_t *c1q, *c2q;                  /* [S * B + 1], [S * B]: a cell's start / 4 in the phase-1 and phase-2 streams */
    int *wg;                              /* [n_wg * 3] */
    int *pw, *pb, *ps;                    /* [n_pieces + 1], [n_pieces], [n_pieces] */
    int *jn;                              /* [n_join * 3] */
} spmv_blocked_plan;
int spmv_blocked_lds_elems(int elem_size);
int spmv_blocked_geometry(int rows, int cols, int elem_size, int bin_rows, spmv_blocked_plan *g);
int spmv_blocked_plan_build(int *cnt, int64_t nnz, int auto_choice, spmv_blocked_plan *g);
void spmv_blocked_plan_free(spmv_blocked_plan *g);
/* auto_choice: decline (-2) a matrix the plan does not suit (columns local to the rows, one bin far heavier than the rest) */
int spmv_blocked_build_device(const int *d_rowptr, const int *d_col, const void *d_val, int rows, int cols, int elem_size,
                              int64_t nnz, int auto_choice, void *stream, spmv_dev_blocked *out);
void spmv_blocked_free(spmv_dev_blocked *d);
int spmvk_blocked(const spmv_dev_blocked *d, const void *x, void *y, void *stream);

/* HUB (spmv_hub.hip): the CSR stream as it lies -- values untouched -- read in fixed chunks by persistent workgroups that
 * keep the most referenced ("hub") columns of x in LDS; a remapped copy of the column indices says per entry hub slot or
 * column, and where a row ends.  For power-law graphs, where a few thousand columns take most of the gathers. */
#define SPMV_HUB_CHUNK 1024        /* entries per chunk (a wave's unit of work)                                  */
#define SPMV_HUB_WAVES 16          /* waves per workgroup: one workgroup per CU                                  */
#define SPMV_HUB_FLAG_HUB 0x80000000u /* col2: bits 0..28 = slot of the LDS table, else the 0-based column       */
#define SPMV_HUB_FLAG_END 0x40000000u /* col2: the entry is the last of its row                                   */
#define SPMV_HUB_FLAG_TBL 0x20000000u /* col2 (not HUB): bits 0..28 index the packed table in memory, not x       */
#define SPMV_HUB_TABLE_BYTES (2 << 20) /* default size of that table: what an XCD's 4-MB L2 keeps next to the stream */
typedef struct spmv_hub_plan {
    int K;            /* tier-0 hub columns: the LDS table (<= spmv_hub_lds_slots)                                */
    int K2;           /* tier-1 hub columns: the dense table in memory                                            */
    int64_t covered;  /* entries whose column is a tier-0 hub                                                      */
    int64_t covered2; /* entries whose column is a tier-1 hub                                                      */
    int cols_used;    /* columns referenced at all                                                                 */
    int CH, n_chunks; /* entries per chunk, chunks                                                                 */
    int n_nzrows, n_empty, n_fix, identity;
    int *hubcol;      /* [K + K2] 0-based columns, ascending within each tier                                      */
    int *slot_of;     /* [cols] slot ([0, K) tier 0, [K, K + K2) tier 1) or -1                                     */
    int *rank0;       /* [n_chunks + 1]                                                                            */
    int *nzrow;       /* [n_nzrows]                                                                                */
    int *erow;        /* [n_empty]                                                                                 */
    int *fix;         /* [n_fix * 3] row, first chunk, chunk of the row's end                                      */
} spmv_hub_plan;
int spmv_hub_lds_slots(int elem_size);
int spmv_hub_select(const int *colcount, int cols, int kmax, int kmax2, int min_count, spmv_hub_plan *g);
int spmv_hub_rows(const int *rowptr, int base, int rows, int ch, spmv_hub_plan *g);
void spmv_hub_plan_free(spmv_hub_plan *g);
typedef struct spmv_dev_hub {
    unsigned int *col2; /* [nnz + SPMV_PAD_ELEMS] remapped columns with flags; the pads point to the table's zero      */
    int *hubcol;        /* [base1 + K2]: the column behind every slot of xh, -1 = none (xh stays 0 there)              */
    void *xh;           /* T[base1 + K2 + 16]: the packed hub entries of x, rewritten by every product: [0, K) the LDS  */
                        /* table, xh[K] = 0 (what the pads multiply with), [base1, base1 + K2) the table in memory     */
    int *rank0, *nzrow, *erow, *fix;
    void *carry;        /* T[n_chunks]: scratch, rewritten by every product                                           */
    const void *val;    /* the CSR copy's values (not owned)                                                          */
    int elem, rows, cols, K, K2, base1, slots, CH, n_chunks, n_nzrows, n_empty, n_fix, identity, n_wg;
    int64_t nnz, covered, covered2;
    size_t bytes;
} spmv_dev_hub;
/* rowptr_host: the block's row pointers on the host, rowptr_host[0] = their base.  0; -2 not suited (auto_choice: too few entries hit a hub) or
 * not representable (2^30 columns or more); -3 out of memory; -1 a HIP error */
int spmv_hub_build_device(const int *d_rowptr, const int *rowptr_host, const int *d_col, const void *d_val, int rows, int cols,
                          int elem_size, int64_t nnz, int auto_choice, void *stream, spmv_dev_hub *out);
void spmv_hub_free(spmv_dev_hub *d);
int spmvk_hub(const spmv_dev_hub *d, const void *x, void *y, void *stream);

struct spmv_hip_matrix;
/* y = A x plus block partials of y . v when the plan can fold them in (STRIPED): returns the number
 * of partials (one per wave range when a workgroup walks every stripe, else one per 256 rows; <= rows + 16), 0 = not folded (run a separate dot), -1 = error */
int spmvd_spmv_dot(struct spmv_hip_matrix *m, const void *x_dev, void *y_dev, void *stream, const void *v_dev,
                   void *dot_part);

/* dotv != NULL: also dot_part[i] = sum of y[r] * dotv[r] over wave range i (NG == 1) or over rows [256 i, 256 i + 256) */
int spmvk_striped(const spmv_dev_striped *t, const void *x, void *y, void *stream, const void *dotv,
                  void *dot_part);
/* d_out[0] = max(colidx), d_out[1] = min(colidx); the caller presets {0, INT_MAX} */
int spmvk_colrange(const int *colidx, int64_t nnz, int *d_out, void *stream);
/* rowptr0[i] = rowstr[i] - rowstr[0] on device (in place allowed) */
int spmvk_rebase(int *rowptr, int n_plus_1, int base, void *stream);
int spmvk_copy_words(void *dst, const void *src, size_t bytes, void *stream);

/* ---- HIP-side helpers for the C host file (spmv_device.hip) ------------------ */
void *spmvd_host_alloc(size_t bytes);            /* pinned host memory */
void spmvd_host_free(void *p);
int spmvd_host_register(void *p, size_t bytes);  /* pin caller memory in place */
void spmvd_host_unregister(void *p);
int spmvd_copy_h2d_async(int device, void *dst_dev, const void *src, size_t bytes);
int spmvd_copy_d2h_async(int device, void *dst, const void *src_dev, size_t bytes);
int spmvd_stream_sync(int device);
/* y = A x in row chunks, every chunk's y on its way to y_host (second stream) while the next chunks compute: 1 done
 * (product and copy complete), 0 the plan does not cut into row chunks (nothing launched), -1 error */
int spmvd_spmv_y_overlapped(struct spmv_hip_matrix *m, const void *x_dev, void *y_dev, void *y_host, int chunks);
/* 0: the automatic plan choice of matrices created BY THE CALLING THREAD from now on skips the re-laid-out plans */
void spmvd_set_relayout(int on);
/* zero-copy: the device-side address of pinned host memory (hipHostMalloc'ed or registered), NULL
 * when it has none; and a word copy kernel on the device's stream whose source may be such memory */
void *spmvd_host_devptr(void *pinned_host);
int spmvd_copy_words_async(int device, void *dst_dev, const void *src_devptr, size_t bytes);

/* ---- misc ------------------------------------------------------------------- */
/* Host threads for the one-time layout conversions (spmv_plan.c): fn(lo, hi, ctx) over [0, n) in
 * contiguous chunks; SPMV_HIP_BUILD_THREADS overrides min(8, cores). */
typedef void (*spmv_range_fn)(int64_t lo, int64_t hi, void *ctx);
int spmv_build_threads(void);
void spmv_parallel_for(int64_t n, int64_t grain, spmv_range_fn fn, void *ctx);

void spmv_set_error(const char *fmt, ...);
double spmv_now(void); /* seconds, CLOCK_MONOTONIC */

#ifdef __cplusplus
}
#endif
#endif
