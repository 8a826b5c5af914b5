/*
 * csrc/spmv_striped.c -- host-side conversion of a CSR row block into the STRIPED layout used
 * by k_striped (plain C, no GPU needed).
 *
 * STRIPED is the 2-D decomposition of y = A*x: the columns are cut into stripes of W columns
 * (W * sizeof(T) = 64 KiB of x), and a workgroup owns ONE stripe for its whole life.  It copies
 * that stripe of x into LDS once and then only streams matrix entries, so the per-nonzero gather
 * of x is an LDS access (as in TILED, spmv_tiled.c) but x is staged S*B times in total instead of
 * once per row block per stripe, and there is no barrier after the staging.  The price is one
 * partial sum per (stripe, row), added up in stripe order by k_stripe_sum.
 *
 * Layout (offsets in entries):
 *
 *   stripe s   holds, row after row, the entries of every row that fall into columns
 *              [s*W, (s+1)*W), each row's entries in their original CSR order;
 *   val[]      T, packed;
 *   col[]      uint16: bits 0..14 = column - 1 - s*W, bit 15 = "last entry of its row in this
 *              stripe".  Every row owns exactly one flagged entry per stripe -- a row with no
 *              nonzero in the stripe gets a dummy (value 0, column 0, flag set) -- so a wave
 *              recovers row numbers by counting flags and needs no row pointers at all;
 *   waves      the rows of a stripe are cut into B*WPB contiguous ranges of about equal entry
 *              counts; wave (s, b, w) streams [wptr[i], wptr[i+1]), i = (s*B + b)*WPB + w, which
 *              is a whole number of sub-windows of 64*EPL entries (pads: value 0, column 0, no
 *              flag) and begins with row wrow[i];
 *   values     inside each sub-window the values are stored lane-interleaved in 16-byte units:
 *              entry e = l*EPL + i sits at (i/G)*(64*G) + l*G + i%G, G = 16/sizeof(T), so that
 *              lane l, which owns entries l*EPL.., reads them with dense 16-byte-per-lane loads
 *              (the 16-bit columns of a lane are contiguous already).
 *
 * Summation order: inside a (row, stripe) segment a fixed tree (4-entry lane sums, then a
 * segmented wave scan), across stripes ascending.  Deterministic, but not the reference's
 * left-to-right order: results agree with native-impl.c within the reordering bound, not bit for
 * bit (TILED and ORDERED are the bit-exact plans).
 */
#include "spmv_internal.h"

#include <stdlib.h>
#include <string.h>

void spmv_striped_free(spmv_striped_host *t)
{
    free(t->val);
    free(t->col);
    free(t->wptr);
    free(t->wrow);
    memset(t, 0, sizeof *t);
}

void spmv_striped_geometry(int rows, int cols, int elem_size, int xbytes, int wpb, int target_wgs, int epl,
                           spmv_striped_host *t)
{
    memset(t, 0, sizeof *t);
    t->elem = elem_size;
    t->EPL = epl == 4 ? 4 : 8;
    t->WPB = wpb;
    int wmax = xbytes / elem_size; /* columns that fit the LDS stripe buffer */
    if (wmax > 32768)
        wmax = 32768; /* 15 column bits */
    if (target_wgs < 1)
        target_wgs = 1;
    /*
     * Widest stripes that fit LDS (fewest partial sums per row), then as many row ranges per
     * stripe as keep S * B within the target: every workgroup is resident at once, no tail round.
     * Narrower stripes that fill the last workgroup slots were measured and lose: on a class D row
     * block S = 125 (250 workgroups) ran 270 us against 250 us for S = 92 (184 workgroups), because
     * every extra stripe adds a row-end marker and a partial sum per row.
     */
    const int s_min = cols > 0 ? (cols + wmax - 1) / wmax : 1;
    int best_w = wmax, best_s = s_min, best_b = target_wgs / s_min > 0 ? target_wgs / s_min : 1;
    if (best_b < 1)
        best_b = 1;
    /* no point in more waves than rows */
    while (best_b > 1 && (int64_t)best_b * wpb > (int64_t)rows)
        --best_b;
    t->W = best_w;
    t->S = best_s;
    t->B = best_b;
}

/*
 * rowstr / colidx / a: the caller's 1-based CSR; rows [r0,r1) are converted.
 * Returns 0, -1 (out of memory), -2 (not representable / not worth it).
 */
int spmv_striped_build(const void *a, const int *rowstr, const int *colidx, int r0, int r1, int cols,
                       int elem_size, int xbytes, int wpb, int target_wgs, int epl, spmv_striped_host *t)
{
    const int rows = r1 - r0;
    spmv_striped_geometry(rows, cols, elem_size, xbytes, wpb, target_wgs, epl, t);
    const int S = t->S, W = t->W, B = t->B, WPB = t->WPB;
    const int64_t SUB = 64 * t->EPL;            /* entries per sub-window */
    const int G = 16 / elem_size;               /* values per 16-byte unit */
    if (rows <= 0 || (int64_t)S * rows > ((int64_t)1 << 28))
        return -2;
    const int64_t nnz = (int64_t)rowstr[r1] - rowstr[r0];

    /* cnt[s*rows + r] = entries of row r in stripe s */
    int32_t *cnt = (int32_t *)calloc((size_t)S * rows, sizeof(int32_t));
    if (!cnt)
        return -1;
    for (int r = 0; r < rows; ++r)
        for (int j = rowstr[r0 + r] - 1; j < rowstr[r0 + r + 1] - 1; ++j) {
            const int c = colidx[j] - 1;
            if (c < 0 || c >= cols) {
                free(cnt);
                return -2;
            }
            cnt[(size_t)(c / W) * rows + r]++;
        }

    const int nw = S * B * WPB;
    t->n_waves = nw;
    t->wptr = (int32_t *)calloc((size_t)nw + 1, sizeof(int32_t));
    t->wrow = (int32_t *)calloc((size_t)nw + 1, sizeof(int32_t));
    /* start[s*rows + r] = position of row r's first entry in stripe s (absolute) */
    int32_t *start = (int32_t *)malloc((size_t)S * rows * sizeof(int32_t));
    if (!t->wptr || !t->wrow || !start) {
        free(cnt), free(start);
        spmv_striped_free(t);
        return -1;
    }

    int64_t total = 0, dummies = 0;
    for (int s = 0; s < S; ++s) {
        const int32_t *cs = cnt + (size_t)s * rows;
        int64_t stripe_entries = 0;
        for (int r = 0; r < rows; ++r)
            stripe_entries += cs[r] > 0 ? cs[r] : 1;
        /* cut the rows into B*WPB ranges of about equal entry counts */
        const int parts = B * WPB;
        int r = 0;
        int64_t done = 0;
        for (int p = 0; p < parts; ++p) {
            const int wi = s * parts + p;
            total = (total + SUB - 1) / SUB * SUB; /* every wave streams whole sub-windows */
            t->wptr[wi] = (int32_t)total;
            t->wrow[wi] = r;
            const int64_t target = stripe_entries * (p + 1) / parts;
            /* leave at least one row for every remaining part when possible */
            while (r < rows && (done < target || p == parts - 1)) {
                if (p < parts - 1 && rows - r <= parts - 1 - p)
                    break;
                const int c = cs[r] > 0 ? cs[r] : 1;
                dummies += cs[r] == 0;
                start[(size_t)s * rows + r] = (int32_t)total;
                total += c;
                done += c;
                ++r;
                if (total > 0x7fffffff - 65536) {
                    free(cnt), free(start);
                    spmv_striped_free(t);
                    return -2;
                }
            }
        }
        if (r != rows) { /* cannot happen: the last part takes everything left */
            free(cnt), free(start);
            spmv_striped_free(t);
            return -2;
        }
    }
    total = (total + SUB - 1) / SUB * SUB;
    t->wptr[nw] = (int32_t)total;
    t->wrow[nw] = rows;
    t->n_packed = total;
    t->n_dummies = dummies;
    if (dummies * 8 > nnz) { /* more than 1/8 of the stream would be filler: the matrix is too
                                sparse per stripe for this plan */
        free(cnt), free(start);
        spmv_striped_free(t);
        return -2;
    }

    const size_t n_alloc = (size_t)total + SPMV_TILE_PAD;
    t->val = calloc(n_alloc, (size_t)elem_size);
    t->col = (uint16_t *)calloc(n_alloc, sizeof(uint16_t));
    if (!t->val || !t->col) {
        free(cnt), free(start);
        spmv_striped_free(t);
        return -1;
    }
    /* scatter in CSR order; `fill` counts the entries already placed per (stripe, row) */
    int32_t *fill = (int32_t *)calloc((size_t)S * rows, sizeof(int32_t));
    if (!fill) {
        free(cnt), free(start);
        spmv_striped_free(t);
        return -1;
    }
    for (int r = 0; r < rows; ++r)
        for (int j = rowstr[r0 + r] - 1; j < rowstr[r0 + r + 1] - 1; ++j) {
            const int c = colidx[j] - 1;
            const int s = c / W;
            const size_t k = (size_t)s * rows + r;
            const int64_t dst = (int64_t)start[k] + fill[k]++;
            uint16_t cc = (uint16_t)(c - s * W);
            if (fill[k] == cnt[k])
                cc |= 0x8000u; /* last entry of this row in this stripe */
            t->col[dst] = cc;
            /* values: lane-interleaved in 16-byte units inside each sub-window, so that lane l
             * (entries l*EPL .. l*EPL+EPL-1) fetches them with dense 16-byte-per-lane loads */
            const int64_t blk = dst / SUB * SUB, rr = dst % SUB;
            const int64_t l = rr / t->EPL, i = rr % t->EPL;
            const int64_t vdst = blk + (i / G) * (64 * G) + l * G + (i % G);
            if (elem_size == 8)
                ((double *)t->val)[vdst] = ((const double *)a)[j];
            else
                ((float *)t->val)[vdst] = ((const float *)a)[j];
        }
    for (int s = 0; s < S; ++s)
        for (int r = 0; r < rows; ++r)
            if (cnt[(size_t)s * rows + r] == 0)
                t->col[start[(size_t)s * rows + r]] = 0x8000u; /* dummy: value 0, flag only */
    free(cnt), free(start), free(fill);
    return 0;
}
