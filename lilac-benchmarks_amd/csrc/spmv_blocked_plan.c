/*
 * csrc/spmv_blocked_plan.c -- the host side of the BLOCKED plan (spmv_blocked.hip): every DECISION of the layout -- stripes and
 * bins, where each (stripe, bin) cell lies in the two streams, phase 1's workgroups, phase 2's pieces and which bins are
 * cut -- from the matrix's dimensions and one number per cell (its entries, counted on the device).  Plain C so that the
 * CPU tests can walk the result (tests/test_blocked_layout.py) without a GPU; the data movement is the .hip file's.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_internal.h"

/* elements of x a workgroup of phase 1 keeps in LDS; the last one is the zero column the pads point to: 160 KB of doubles,
 * 15-bit columns of floats */
int spmv_blocked_lds_elems(int elem_size) { return elem_size == 8 ? 20480 : 32768; }

/*
 * Rows per bin.  Phase 2 runs one wave per bin, four waves and 4 RB 8 bytes of LDS per workgroup: up to 2560 rows two
 * workgroups share a CU -- 2048 waves at a time on 256 CUs.  A matrix with a few more bins than that pays a second, nearly
 * empty round (SparseBench 170^3 in bins of 2048 rows: 2399 bins, 193 against 176 us), so the bins are sized to fill whole
 * rounds; never under 1024 rows (smaller bins make smaller cells: SparseBench 120^3 63 / 70 / 75 us with 1024 / 768 / 512).
 */
static int choose_bin_rows(int rows)
{
    /* 1946 of the 2048 slots of a round: one bin in twenty may be cut (more and the builder declines) without its pieces
     * spilling into another round */
    const int64_t slots = 1946, rb_max = SPMV_BLOCKED_MAX_BIN_ROWS;
    const int64_t rounds = (rows + slots * rb_max - 1) / (slots * rb_max);
    int64_t rb = (rows + slots * rounds - 1) / (slots * rounds);
    rb = (rb + 255) / 256 * 256;
    return (int)(rb < 1024 ? 1024 : rb > rb_max ? rb_max : rb);
}

/* Stripes: as few as the LDS allows, in multiples of 256 (one per CU and round); narrow matrices get stripes of 64 columns
 * or more.  0, or -2 when the matrix is not representable (more stripes than the builder's counters, more than 2^24 cells) */
int spmv_blocked_geometry(int rows, int cols, int elem_size, int bin_rows, spmv_blocked_plan *g)
{
    memset(g, 0, sizeof *g);
    if (rows <= 0 || cols <= 0 || (elem_size != 8 && elem_size != 4))
        return -2;
    const int E = 16 / elem_size, W = spmv_blocked_lds_elems(elem_size);
    int S, wcols;
    if (cols <= 256 * 64) {
        wcols = 64;
        S = (cols + wcols - 1) / wcols;
    } else {
        int k = 1;
        while ((int64_t)256 * k * (W - E) < cols)
            ++k;
        S = 256 * k;
        wcols = (cols + S - 1) / S;
        wcols = (wcols + E - 1) / E * E;
        S = (cols + wcols - 1) / wcols;
    }
    const int RB = bin_rows > 0 ? bin_rows : choose_bin_rows(rows);
    if (RB < 64 || RB > SPMV_BLOCKED_MAX_BIN_ROWS)
        return -2;
    const int B = (rows + RB - 1) / RB;
    if (S > SPMV_BLOCKED_MAX_STRIPES || (int64_t)S * B > ((int64_t)1 << 24))
        return -2;
    g->S = S, g->wcols = wcols, g->W = W, g->RB = RB, g->B = B;
    return 0;
}

void spmv_blocked_plan_free(spmv_blocked_plan *g)
{
    free(g->c1q), free(g->c2q), free(g->wg), free(g->pw), free(g->pb), free(g->ps), free(g->jn);
    g->c1q = g->c2q = NULL;
    g->wg = g->pw = g->pb = g->ps = g->jn = NULL;
}

/*
 * The two orders, from cnt[s * B + b] = entries of cell (s, b); on return cnt holds the PADDED sizes (multiples of four).
 *   phase 1: stripe-major, cells padded to fours, a stripe to the window (256 entries); c1q[id] = the cell's start / 4,
 *            c1q[S * B] = n1 / 4.  What lies between a stripe's last cell and the window boundary are pads whose products
 *            (0 x 0) go to the spare group behind phase 2's stream, not into a bin (when they went into the stripe's last
 *            bin, that bin held 3.5 x the mean and phase 2 waited for its wave: SparseBench 170^3 176 -> 141 us);
 *   phase 2: bin-major with the same padded cells; c2q[id] = the cell's start / 4.  A bin is one PIECE -- one wave -- unless
 *            it holds more than its share (dense rows): then it is cut, between stripes, into pieces of about the mean bin,
 *            each padded to the window (slots nobody writes: they stay zero), each with a slot for its partial sums, which
 *            k_blk_join adds in order.  More pieces than wave slots are no harm: workgroups are handed out in order and a
 *            light bin's wave is gone in no time.
 *   phase 1's workgroups: one per stripe; a stripe with more than 5/4 of the mean share is cut.
 * 0; -2: a stream beyond 2^31 entries, or (auto_choice) the plan does not suit the matrix; -3: no memory; -1: the counts do
 * not add up to nnz.
 */
int spmv_blocked_plan_build(int *cnt, int64_t nnz, int auto_choice, spmv_blocked_plan *g)
{
    const int S = g->S, B = g->B;
    const size_t ncells = (size_t)S * (size_t)B;
    const int64_t WIN = SPMV_BLOCKED_WIN;
    int rc = 0;
    int64_t *stripe_w0 = (int64_t *)malloc(((size_t)S + 1) * sizeof(int64_t));
    int64_t *bin_e = (int64_t *)calloc((size_t)B, sizeof(int64_t));
    g->c1q = (uint32_t *)malloc((ncells + 1) * sizeof(uint32_t));
    g->c2q = (uint32_t *)malloc(ncells * sizeof(uint32_t));
    g->jn = (int *)malloc(((size_t)B + 1) * 3 * sizeof(int));
    if (!stripe_w0 || !bin_e || !g->c1q || !g->c2q || !g->jn) {
        rc = -3;
        goto out;
    }
    int64_t p = 0, total = 0, padded_total = 0;
    for (int s = 0; s < S; ++s) {
        stripe_w0[s] = p / WIN;
        for (int b = 0; b < B; ++b) {
            const size_t id = (size_t)s * B + b;
            if (cnt[id] < 0) {
                rc = -1;
                goto out;
            }
            total += cnt[id];
            cnt[id] = (cnt[id] + 3) & ~3; /* to eights (64-byte-aligned runs of products): 146 us against 144 */
            g->c1q[id] = (uint32_t)(p / 4);
            p += cnt[id];
            bin_e[b] += cnt[id];
            padded_total += cnt[id];
        }
        p = (p + WIN - 1) / WIN * WIN;
        if (p > ((int64_t)1 << 31) - 4 * WIN) {
            rc = -2;
            goto out;
        }
    }
    stripe_w0[S] = p / WIN;
    g->c1q[ncells] = (uint32_t)(p / 4);
    g->n1 = p;
    if (total != nnz) { /* the caller reports it */
        rc = -1;
        goto out;
    }
    /* ---- phase 2's pieces; the share of a piece: a quarter above the mean bin */
    {
        const int64_t share = (padded_total / B + 1) * 5 / 4 + 1024;
        size_t cap_pieces = 0;
        for (int b = 0; b < B; ++b)
            cap_pieces += (size_t)(bin_e[b] > share * 6 / 5 ? (bin_e[b] + share - 1) / share : 1);
        g->pw = (int *)malloc((cap_pieces + 1) * sizeof(int));
        g->pb = (int *)malloc((cap_pieces + 1) * sizeof(int));
        g->ps = (int *)malloc((cap_pieces + 1) * sizeof(int));
        if (!g->pw || !g->pb || !g->ps) {
            rc = -3;
            goto out;
        }
        int n_pieces = 0, n_slots = 0, n_join = 0;
        int64_t max_piece = 0;
        p = 0;
        for (int b = 0; b < B; ++b) {
            const int want = (int)(bin_e[b] > share * 6 / 5 ? (bin_e[b] + share - 1) / share : 1);
            const int64_t per = (bin_e[b] + want - 1) / want;
            int opened = 1;
            int64_t acc = 0;
            g->pw[n_pieces] = (int)(p / WIN), g->pb[n_pieces] = b, g->ps[n_pieces] = want > 1 ? n_slots : -1;
            if (want > 1)
                g->jn[3 * n_join] = b, g->jn[3 * n_join + 1] = n_slots;
            for (int s = 0; s < S; ++s) {
                const size_t id = (size_t)s * B + b;
                if (acc >= per && opened < want && cnt[id]) { /* the next piece of this bin starts with this stripe */
                    max_piece = acc > max_piece ? acc : max_piece;
                    p = (p + WIN - 1) / WIN * WIN;
                    ++n_pieces, ++opened, acc = 0;
                    g->pw[n_pieces] = (int)(p / WIN), g->pb[n_pieces] = b, g->ps[n_pieces] = n_slots + opened - 1;
                }
                g->c2q[id] = (uint32_t)(p / 4);
                p += cnt[id];
                acc += cnt[id];
            }
            max_piece = acc > max_piece ? acc : max_piece;
            if (want > 1) {
                g->jn[3 * n_join + 2] = opened;
                n_slots += opened;
                ++n_join;
            }
            ++n_pieces;
            p = (p + WIN - 1) / WIN * WIN;
            if (p > ((int64_t)1 << 31) - 4 * WIN) {
                rc = -2;
                goto out;
            }
        }
        g->pw[n_pieces] = (int)(p / WIN);
        g->n2 = p;
        g->n_pieces = n_pieces, g->n_slots = n_slots, g->n_join = n_join;
        /*
         * An automatic choice is taken back when the plan cannot pay off (spmv_plan_choose nominates by shape alone):
         *  - the columns are LOCAL to the rows (banded): the CSR kernels' gathers of x hit in cache and nothing is gained for
         *    the products' extra trip through memory.  Seen from here: far fewer cells hold entries than scattered columns
         *    would fill (a cell of mean m entries is empty with probability exp(-m));
         *  - a piece that cannot be cut -- one cell -- is several times the share: phase 2 lasts as long as its heaviest
         *    piece's wave, and a cell that crowded holds runs of one row's products, which the LDS adds one after the
         *    other (R-MAT 2^21: a cell of 106 000 entries where the mean BIN has 16 000, 62 % of all entries follow one of
         *    their own row; BLOCKED by name 0.40-0.53 ms, STREAM 0.26);
         *  - more than one bin in twenty is cut: the rows are uneven throughout, not here and there (the same graph without
         *    its 16 384 most popular columns: no crowded cell any more, but 232 of 2048 bins cut, 449 us where the CSR
         *    kernel's gathers are no worse than on the whole graph).
         */
        if (auto_choice) {
            size_t nonempty = 0;
            for (size_t id = 0; id < ncells; ++id)
                nonempty += cnt[id] != 0;
            const double mean_cell = (double)nnz / (double)ncells;
            const double expected = (double)ncells * (1.0 - exp(-mean_cell));
            if ((double)nonempty < 0.6 * expected || max_piece > 3 * share || (int64_t)n_join * 20 > B) {
                rc = -2;
                goto out;
            }
        }
    }
    /* ---- phase 1's workgroups */
    {
        const int64_t slots = S > 256 ? S : 256;
        int64_t target = (stripe_w0[S] + slots - 1) / slots;
        target = target * 5 / 4 > 0 ? target * 5 / 4 : 1;
        size_t cap = 0;
        for (int s = 0; s < S; ++s)
            cap += (size_t)((stripe_w0[s + 1] - stripe_w0[s] + target - 1) / target);
        g->wg = (int *)malloc((cap + 1) * 3 * sizeof(int));
        if (!g->wg) {
            rc = -3;
            goto out;
        }
        int n_wg = 0;
        for (int s = 0; s < S; ++s) {
            const int64_t a = stripe_w0[s], e = stripe_w0[s + 1];
            const int parts = (int)((e - a + target - 1) / target);
            for (int q = 0; q < parts; ++q) {
                g->wg[3 * n_wg] = s;
                g->wg[3 * n_wg + 1] = (int)(a + (e - a) * q / parts);
                g->wg[3 * n_wg + 2] = (int)(a + (e - a) * (q + 1) / parts);
                ++n_wg;
            }
        }
        g->n_wg = n_wg;
    }
out:
    free(stripe_w0), free(bin_e);
    if (rc)
        spmv_blocked_plan_free(g);
    return rc;
}
