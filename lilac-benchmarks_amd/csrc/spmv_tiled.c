/*
 * csrc/spmv_tiled.c -- host-side conversion of a CSR row block into the TILED
 * layout used by k_tiled (plain C, no GPU needed).
 *
 * Why: with columns scattered over the whole of x (NPB CG: outer products of
 * random sparse vectors, /root/reference/NPB3.3.1/CG/cg.f:710-719), every a[j]
 * costs one 8-byte gather that pulls a whole cache line from L2 into the CU.
 * Measured on MI355X, class C: 224 us per SpMV against 75 us for the same row
 * lengths with local columns (profiles/r01_kbench*.log) -- the gather, not
 * HBM, is the bound.  k_tiled keeps a stripe of x in LDS instead, where a
 * random 8-byte read costs a bank access rather than a cache line.
 *
 * Layout (all offsets in elements):
 *
 *   stripes   S = ceil(cols / W) column ranges of W columns (W * sizeof(T) = 32 KiB of LDS,
 *             double-buffered by the kernel)
 *   blocks    B = ceil(rows / (WPB * RW)) row blocks; a block is one workgroup of WPB waves,
 *             wave w of block b owns the RW (<= 64) consecutive rows
 *             [(b*WPB + w)*RW, +RW): one row per lane, accumulator in a register
 *   chunk     (g, s) = the nonzeros of wave-group g = b*WPB + w that fall in stripe s,
 *             stored contiguously, row after row, each row's entries in their original
 *             CSR order; chunks are ordered g-major, s-minor, so a wave streams linearly
 *             through [wptr[g*S], wptr[(g+1)*S]) while the workgroup walks the stripes;
 *             every chunk starts on a multiple of 8 elements (pads: val = 0, col = 0)
 *   val[]     T, packed
 *   col[]     uint16: column - 1 - s*W   (local to the stripe)
 *   wptr[]    int32 [n_groups*S + 1]: start of each chunk
 *   seg[]     uint16 [n_groups*S*(RW+1)]: seg[c*(RW+1) + r] = offset of row r's first entry
 *             inside chunk c; seg[..][RW] = entries in the chunk (<= 65535)
 *
 * Summation order inside a row = stripe by stripe, original order within a
 * stripe.  For rows whose columns are sorted (every NPB matrix, cg.f:840-865)
 * that IS the reference's left-to-right order, and since k_tiled rounds each
 * product before adding it, its results are bit-identical to
 * libspmv/native-impl.c:1-12 on such matrices.
 */
#include "spmv_internal.h"

#include <stdlib.h>
#include <string.h>

/* geometry knobs (experiments through spmv_hip_set_tuning keys 2..4) */
static int g_tile_wpb = SPMV_TILE_WPB;
static int g_tile_xbytes = SPMV_TILE_LDS_BYTES;
static int g_tile_blocks = 0; /* 0: the caller's target */

void spmv_tiled_set_wpb(int wpb)
{
    if (wpb == SPMV_TILE_WPB)
        g_tile_wpb = wpb;
}

void spmv_tiled_set_xbytes(int bytes)
{
    if (bytes == SPMV_TILE_LDS_BYTES)
        g_tile_xbytes = bytes;
}

void spmv_tiled_set_blocks(int blocks) { g_tile_blocks = blocks > 0 ? blocks : 0; }

void spmv_tiled_free(spmv_tiled_host *t)
{
    free(t->val);
    free(t->col);
    free(t->wptr);
    free(t->seg);
    memset(t, 0, sizeof *t);
}

int spmv_tiled_geometry(int rows, int cols, int elem_size, int target_blocks, spmv_tiled_host *t)
{
    memset(t, 0, sizeof *t);
    t->elem = elem_size;
    t->W = g_tile_xbytes / elem_size;
    t->S = cols > 0 ? (cols + t->W - 1) / t->W : 1;
    t->WPB = g_tile_wpb;
    if (g_tile_blocks > 0)
        target_blocks = g_tile_blocks;
    if (target_blocks < 1)
        target_blocks = 1;
    long per_wave = ((long)rows + (long)t->WPB * target_blocks - 1) / ((long)t->WPB * target_blocks);
    if (per_wave < 1)
        per_wave = 1;
    if (per_wave > 64)
        per_wave = 64;
    t->RW = (int)per_wave;
    const int rb = t->RW * t->WPB;
    t->B = rows > 0 ? (rows + rb - 1) / rb : 0;
    t->n_groups = t->B * t->WPB;
    return 0;
}

/*
 * rowstr / colidx / a: the caller's 1-based CSR; rows [r0,r1) are converted.
 * Returns 0, -1 (out of memory), -2 (not representable: a chunk exceeds 65535
 * entries or the packed size exceeds int32) -- the caller then keeps another plan.
 */
int spmv_tiled_build(const void *a, const int *rowstr, const int *colidx, int r0, int r1, int cols,
                     int elem_size, int target_blocks, spmv_tiled_host *t)
{
    const int rows = r1 - r0;
    spmv_tiled_geometry(rows, cols, elem_size, target_blocks, t);
    const int S = t->S, RW = t->RW, W = t->W;
    const int64_t nchunks = (int64_t)t->n_groups * S;
    if (nchunks * (RW + 1) > (int64_t)1 << 31)
        return -2;

    t->wptr = (int32_t *)calloc((size_t)nchunks + 1, sizeof(int32_t));
    t->seg = (uint16_t *)calloc((size_t)(nchunks * (RW + 1)) + 1, sizeof(uint16_t));
    int32_t *cnt = (int32_t *)calloc((size_t)RW * S + 1, sizeof(int32_t)); /* one group at a time */
    if (!t->wptr || !t->seg || !cnt) {
        free(cnt);
        spmv_tiled_free(t);
        return -1;
    }

    /* pass 1: per (row, stripe) counts -> seg offsets, chunk sizes -> wptr */
    int64_t total = 0;
    for (int g = 0; g < t->n_groups; ++g) {
        const int gr0 = g * RW; /* first local row of the group */
        memset(cnt, 0, (size_t)RW * S * sizeof(int32_t));
        for (int r = 0; r < RW; ++r) {
            const int row = gr0 + r;
            if (row >= rows)
                break;
            for (int j = rowstr[r0 + row] - 1; j < rowstr[r0 + row + 1] - 1; ++j) {
                const int c = colidx[j] - 1;
                if (c < 0 || c >= cols) {
                    free(cnt);
                    spmv_tiled_free(t);
                    return -2;
                }
                cnt[(c / W) * RW + r]++;
            }
        }
        for (int s = 0; s < S; ++s) {
            const int64_t c = (int64_t)g * S + s;
            uint16_t *sg = t->seg + c * (RW + 1);
            int64_t run = 0;
            for (int r = 0; r < RW; ++r) {
                sg[r] = (uint16_t)run;
                run += cnt[s * RW + r];
                if (run > 65535) {
                    free(cnt);
                    spmv_tiled_free(t);
                    return -2;
                }
            }
            sg[RW] = (uint16_t)run;
            t->wptr[c] = (int32_t)total;
            total += (run + 7) & ~(int64_t)7;
            if (total > 0x7fffffff - 4096) {
                free(cnt);
                spmv_tiled_free(t);
                return -2;
            }
        }
    }
    t->wptr[nchunks] = (int32_t)total;
    t->n_packed = total;
    for (int64_t c = 0; c < nchunks; ++c)
        t->nonempty_chunks += t->seg[c * (RW + 1) + RW] != 0;

    const size_t n_alloc = (size_t)total + SPMV_TILE_PAD;
    t->val = calloc(n_alloc, (size_t)elem_size);
    t->col = (uint16_t *)calloc(n_alloc, sizeof(uint16_t));
    if (!t->val || !t->col) {
        free(cnt);
        spmv_tiled_free(t);
        return -1;
    }

    /* pass 2: scatter, keeping the CSR order inside every (row, stripe) segment */
    for (int g = 0; g < t->n_groups; ++g) {
        const int gr0 = g * RW;
        memset(cnt, 0, (size_t)RW * S * sizeof(int32_t)); /* now: entries already placed */
        for (int r = 0; r < RW; ++r) {
            const int row = gr0 + r;
            if (row >= rows)
                break;
            for (int j = rowstr[r0 + row] - 1; j < rowstr[r0 + row + 1] - 1; ++j) {
                const int c = colidx[j] - 1;
                const int s = c / W;
                const int64_t ch = (int64_t)g * S + s;
                const int64_t dst = (int64_t)t->wptr[ch] + t->seg[ch * (RW + 1) + r] + cnt[s * RW + r]++;
                t->col[dst] = (uint16_t)(c - s * W);
                if (elem_size == 8)
                    ((double *)t->val)[dst] = ((const double *)a)[j];
                else
                    ((float *)t->val)[dst] = ((const float *)a)[j];
            }
        }
    }
    free(cnt);
    return 0;
}
