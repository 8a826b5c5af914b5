/*
 * csrc/spmv_lanes.c -- host-side conversion of a CSR row block into the LANES layout used by
 * k_lanes (plain C, no GPU needed).
 *
 * LANES is the STRIPED decomposition (spmv_striped.c: a workgroup keeps ONE stripe of x in LDS for
 * its whole life) with the row sums taken the other way round.  STRIPED streams a stripe's entries
 * row after row across the lanes and needs a segmented wave scan, flag counting and an LDS ring per
 * 256 entries -- about 150 vector instructions, which on class C costs as much time as the memory
 * traffic.  Here every LANE owns whole (row, stripe) segments: it multiplies and adds its own
 * entries front to back (one FMA per entry), and when it meets the last entry of a segment it stores
 * the sum and starts the next one.  No cross-lane operation, no LDS besides x.
 *
 * Layout of one stripe (offsets in entries):
 *
 *   waves      the rows are cut into B*WPB contiguous ranges of about equal entry counts, as in
 *              STRIPED;
 *   groups     inside a wave's range the segments that are not empty are ordered by decreasing
 *              length (ties: by row) and taken 64 at a time; the longest of a group goes to the lane
 *              that has the fewest entries so far, the next to the next, ... (LPT per group), so the
 *              lanes' streams stay within about one short segment of each other all the way;
 *   slots      the segment that lane l receives from group g writes partial[wslot[i] + 64 g + l]:
 *              the kernel needs no row numbers, it adds 64 to its slot at every segment end.  Every
 *              lane has a segment in every group but the last one of the wave;
 *   queue      a lane parks its finished sums in a 4-deep LDS queue and the wave writes group g as
 *              one coalesced 64-lane store once all lanes are past it; a lane that gets further
 *              ahead (runs of 1-entry segments) stores directly and skips that group's write-out;
 *              the layout is refused (-2) when a lane could get 32 groups ahead (one row far longer
 *              than a whole group of the others): STRIPED or STREAM take such matrices;
 *   stream     lane l's segments back to back form its stream; all lanes of a wave are padded
 *              (value 0, column 0, no flag -- after the lane's last segment, where the sum is never
 *              stored) to the longest, rounded up to 4.  Entries 4q..4q+3 of all 64 lanes make
 *              sub-window q = 256 entries stored exactly like a STRIPED sub-window: lane l's four
 *              16-bit columns contiguous, values lane-interleaved in 16-byte units;
 *   col[]      uint16: bits 0..14 = column - 1 - s*W, bit 15 = last entry of its segment;
 *   islot      islot[s*rows + r] is the slot of (stripe s, row r), -1 when the row has nothing in
 *              the stripe (no filler entries exist).  k_lane_sum adds a row's slots in stripe order.
 *
 * Summation order: inside a segment front to back (fused multiply-add), across stripes ascending.
 * Deterministic; not the reference's order when a row spans stripes (TILED and ORDERED are the
 * bit-exact plans).
 */
#include "spmv_internal.h"

#include <stdlib.h>
#include <string.h>

void spmv_lanes_free(spmv_lanes_host *t)
{
    free(t->val);
    free(t->col);
    free(t->wptr);
    free(t->wslot);
    free(t->islot);
    memset(t, 0, sizeof *t);
}

typedef struct lanes_ctx {
    const void *a;
    const int *rowstr, *colidx;
    int r0, rows, cols, elem, S, W, B, WPB, G;
    int32_t *cnt;    /* [S][rows] entries of row r in stripe s                                   */
    int32_t *estart; /* [S][rows] first stream entry of the segment inside its lane              */
    uint8_t *lane;   /* [S][rows] lane that owns the segment                                     */
    int32_t *wave;   /* [S][rows] wave (global index) of the segment                             */
    int64_t *sbase;  /* [S+1] first entry of each stripe                                          */
    int64_t *sslot;  /* [S+1] first slot of each stripe                                           */
    int64_t *wlen;   /* [nw]  entries per wave (multiple of 256)                                  */
    int32_t *wnz;    /* [nw]  non-empty segments per wave                                         */
    int32_t *wrel;   /* [nw]  first entry of the wave relative to its stripe                      */
    spmv_lanes_host *t;
    volatile int bad;
} lanes_ctx;

static void lanes_count(int64_t lo, int64_t hi, void *vc)
{
    lanes_ctx *c = (lanes_ctx *)vc;
    for (int64_t r = lo; r < hi; ++r)
        for (int j = c->rowstr[c->r0 + r] - 1; j < c->rowstr[c->r0 + r + 1] - 1; ++j) {
            const int col = c->colidx[j] - 1;
            if (col < 0 || col >= c->cols) {
                c->bad = 1;
                return;
            }
            c->cnt[(size_t)(col / c->W) * c->rows + r]++;
        }
}

typedef struct seg_key {
    int32_t len, row;
} seg_key;

static int seg_cmp(const void *pa, const void *pb)
{
    const seg_key *a = (const seg_key *)pa, *b = (const seg_key *)pb;
    if (a->len != b->len)
        return a->len > b->len ? -1 : 1; /* longest first */
    return a->row < b->row ? -1 : (a->row > b->row);
}

/* per stripe: wave ranges, ranks, lanes, stream offsets (all relative to the stripe) */
static void lanes_layout(int64_t lo, int64_t hi, void *vc)
{
    lanes_ctx *c = (lanes_ctx *)vc;
    const int rows = c->rows, parts = c->B * c->WPB;
    seg_key *key = (seg_key *)malloc((size_t)(rows > 0 ? rows : 1) * sizeof(seg_key));
    int64_t *tdone = (int64_t *)malloc(((size_t)rows / 64 + 2) * sizeof(int64_t)); /* per group of a wave */
    if (!key || !tdone) {
        free(key), free(tdone);
        c->bad = 2;
        return;
    }
    for (int64_t s = lo; s < hi; ++s) {
        const int32_t *cs = c->cnt + (size_t)s * rows;
        int64_t stripe_entries = 0;
        for (int r = 0; r < rows; ++r)
            stripe_entries += cs[r];
        int r = 0;
        int64_t done = 0, total = 0, slots = 0;
        for (int p = 0; p < parts; ++p) {
            const int wi = (int)s * parts + p;
            const int ra = r;
            const int64_t target = stripe_entries * (p + 1) / parts;
            while (r < rows && (done < target || p == parts - 1)) {
                if (p < parts - 1 && rows - r <= parts - 1 - p)
                    break; /* leave a row for every remaining wave when possible */
                done += cs[r];
                ++r;
            }
            int n = 0;
            for (int q = ra; q < r; ++q)
                if (cs[q] > 0) {
                    key[n].len = cs[q];
                    key[n].row = q;
                    ++n;
                }
            qsort(key, (size_t)n, sizeof(seg_key), seg_cmp);
            int64_t fill[64];
            int order[64];
            memset(fill, 0, sizeof fill);
            for (int k0 = 0, g = 0; k0 < n; k0 += 64, ++g) {
                const int m = n - k0 < 64 ? n - k0 : 64;
                /* lanes by entries so far (ties: by lane): insertion sort of 64 keys */
                for (int l = 0; l < 64; ++l) {
                    int j = l;
                    while (j > 0 && fill[order[j - 1]] > fill[l]) {
                        order[j] = order[j - 1];
                        --j;
                    }
                    order[j] = l;
                }
                int64_t group_done = 0;
                for (int i = 0; i < m; ++i) {
                    const int l = order[i]; /* longest segment of the group -> emptiest lane */
                    const size_t at = (size_t)s * rows + key[k0 + i].row;
                    c->estart[at] = (int32_t)fill[l];
                    c->lane[at] = (uint8_t)l;
                    c->wave[at] = wi;
                    c->t->islot[at] = (int32_t)(slots + 64 * (int64_t)g + l); /* relative to the stripe for now */
                    fill[l] += key[k0 + i].len;
                    group_done = fill[l] > group_done ? fill[l] : group_done;
                    /* how many groups is this lane ahead of the wave's write-out when it finishes the
                     * segment?  Groups leave in order, group g' right after entry step tdone[g'].  Up to
                     * SPMV_LANES_QUEUE - 1 the sum waits in the lane's LDS queue, beyond that the
                     * kernel stores it directly and remembers so in a 32-bit mask: 32 is the limit. */
                    /* write-outs happen between sub-windows: group g' has left when the sub-window
                     * in which its last lane finished lies before the one this segment ends in */
                    const int64_t qw = (fill[l] - 1) / 4;
                    int lo = 0, hi = g; /* first g' in [0, g) not yet written out at that point */
                    while (lo < hi) {
                        const int mid = (lo + hi) / 2;
                        if ((tdone[mid] - 1) / 4 < qw)
                            lo = mid + 1;
                        else
                            hi = mid;
                    }
                    if (g - lo >= 32)
                        c->bad = 3;
                }
                tdone[g] = group_done;
            }
            n = (n + 63) / 64 * 64; /* slots are handed out in whole groups */
            int64_t T = 0;
            for (int l = 0; l < 64; ++l)
                T = fill[l] > T ? fill[l] : T;
            T = (T + 3) / 4 * 4;
            c->wrel[wi] = (int32_t)total;
            c->wlen[wi] = T * 64;
            c->wnz[wi] = n;
            total += T * 64;
            slots += n;
            if (total > 0x7fffffff - 65536 || c->bad) {
                c->bad = c->bad ? c->bad : 3;
                free(key), free(tdone);
                return;
            }
        }
        c->sbase[s + 1] = total;
        c->sslot[s + 1] = slots;
    }
    free(key), free(tdone);
}

static void lanes_scatter(int64_t lo, int64_t hi, void *vc)
{
    lanes_ctx *c = (lanes_ctx *)vc;
    const int rows = c->rows, S = c->S, W = c->W, G = c->G;
    int32_t *fill = (int32_t *)malloc((size_t)S * sizeof(int32_t));
    if (!fill) {
        c->bad = 2;
        return;
    }
    uint16_t *tcol = c->t->col;
    for (int64_t r = lo; r < hi; ++r) {
        const int jb = c->rowstr[c->r0 + r] - 1, je = c->rowstr[c->r0 + r + 1] - 1;
        if (jb == je)
            continue;
        memset(fill, 0, (size_t)S * sizeof(int32_t));
        for (int j = jb; j < je; ++j) {
            const int col = c->colidx[j] - 1;
            const int s = col / W;
            const size_t k = (size_t)s * rows + r;
            const int64_t e = (int64_t)c->estart[k] + fill[s]++; /* position in the lane's stream */
            const int64_t wbase = (int64_t)c->t->wptr[c->wave[k]];
            const int64_t q = e >> 2, i = e & 3, l = c->lane[k];
            uint16_t cc = (uint16_t)(col - s * W);
            if (fill[s] == c->cnt[k])
                cc |= 0x8000u; /* last entry of the segment */
            tcol[wbase + q * 256 + l * 4 + i] = cc;
            const int64_t vdst = wbase + q * 256 + (i / G) * (64 * G) + l * G + (i % G);
            if (c->elem == 8)
                ((double *)c->t->val)[vdst] = ((const double *)c->a)[j];
            else
                ((float *)c->t->val)[vdst] = ((const float *)c->a)[j];
        }
    }
    free(fill);
}

/*
 * rowstr / colidx / a: the caller's 1-based CSR; rows [r0,r1) are converted.
 * Returns 0, -1 (out of memory), -2 (not representable / not worth it).
 * The result does not depend on the number of threads.
 */
int spmv_lanes_build(const void *a, const int *rowstr, const int *colidx, int r0, int r1, int cols,
                     int elem_size, int xbytes, int wpb, int target_wgs, spmv_lanes_host *t)
{
    const int rows = r1 - r0;
    spmv_striped_host g;
    spmv_striped_geometry(rows, cols, elem_size, xbytes, wpb, target_wgs, 4, &g);
    memset(t, 0, sizeof *t);
    t->elem = elem_size, t->W = g.W, t->S = g.S, t->WPB = g.WPB, t->B = g.B;
    /* a wave wants at least 64 segments, one per lane: fewer row ranges for a short matrix */
    while (t->B > 1 && (int64_t)t->B * t->WPB * 64 > (int64_t)rows)
        --t->B;
    const int S = t->S;
    if (rows <= 0 || (int64_t)S * rows > ((int64_t)1 << 28))
        return -2;
    const int64_t nnz = (int64_t)rowstr[r1] - rowstr[r0];
    const int nw = S * t->B * t->WPB;
    t->n_waves = nw;

    lanes_ctx c;
    memset(&c, 0, sizeof c);
    c.a = a, c.rowstr = rowstr, c.colidx = colidx;
    c.r0 = r0, c.rows = rows, c.cols = cols, c.elem = elem_size;
    c.S = S, c.W = t->W, c.B = t->B, c.WPB = t->WPB, c.G = 16 / elem_size;
    c.t = t;
    int rc = -1;
    const size_t sr = (size_t)S * rows;
    c.cnt = (int32_t *)calloc(sr, sizeof(int32_t));
    c.estart = (int32_t *)malloc(sr * sizeof(int32_t));
    c.lane = (uint8_t *)malloc(sr);
    c.wave = (int32_t *)malloc(sr * sizeof(int32_t));
    c.sbase = (int64_t *)calloc((size_t)S + 1, sizeof(int64_t));
    c.sslot = (int64_t *)calloc((size_t)S + 1, sizeof(int64_t));
    c.wlen = (int64_t *)calloc((size_t)nw + 1, sizeof(int64_t));
    c.wnz = (int32_t *)calloc((size_t)nw + 1, sizeof(int32_t));
    c.wrel = (int32_t *)calloc((size_t)nw + 1, sizeof(int32_t));
    t->wptr = (int32_t *)calloc((size_t)nw + 1, sizeof(int32_t));
    t->wslot = (int32_t *)calloc((size_t)nw + 1, sizeof(int32_t));
    t->islot = (int32_t *)malloc(sr * sizeof(int32_t));
    if (!c.cnt || !c.estart || !c.lane || !c.wave || !c.sbase || !c.sslot || !c.wlen || !c.wnz || !c.wrel ||
        !t->wptr || !t->wslot || !t->islot)
        goto out;
    memset(t->islot, 0xff, sr * sizeof(int32_t)); /* -1: the row has nothing in the stripe */

    spmv_parallel_for(rows, 4096, lanes_count, &c);
    if (c.bad) {
        rc = -2;
        goto out;
    }
    spmv_parallel_for(S, 1, lanes_layout, &c);
    if (c.bad) {
        rc = c.bad == 2 ? -1 : -2;
        goto out;
    }
    {
        int64_t total = 0, slots = 0;
        for (int s = 0; s < S; ++s) {
            const int64_t len = c.sbase[s + 1], ns = c.sslot[s + 1];
            c.sbase[s] = total;
            c.sslot[s] = slots;
            total += len;
            slots += ns;
        }
        c.sbase[S] = total;
        c.sslot[S] = slots;
        if (total > 0x7fffffff - 65536 || slots > 0x7fffffff - 65536) {
            rc = -2;
            goto out;
        }
        const int parts = t->B * t->WPB;
        int64_t slot = 0;
        for (int wi = 0; wi < nw; ++wi) {
            t->wptr[wi] = (int32_t)(c.sbase[wi / parts] + c.wrel[wi]);
            t->wslot[wi] = (int32_t)slot;
            slot += c.wnz[wi];
        }
        t->wptr[nw] = (int32_t)total;
        t->wslot[nw] = (int32_t)slot; /* == slots */
        t->n_packed = total;
        t->n_slots = slots;
        t->n_pad = total - nnz;
        /* islot: stripe-relative -> absolute */
        for (int s = 1; s < S; ++s) {
            const int32_t add = (int32_t)c.sslot[s];
            int32_t *is = t->islot + (size_t)s * rows;
            for (int r = 0; r < rows; ++r)
                if (is[r] >= 0)
                    is[r] += add;
        }
    }
    if (t->n_pad > 4 * nnz + ((int64_t)1 << 22)) { /* lanes hopelessly uneven (one row is most of the
                                                     matrix): the caller's own limit is n_pad vs nnz */
        rc = -2;
        goto out;
    }
    {
        const size_t n_alloc = (size_t)t->n_packed + SPMV_TILE_PAD;
        t->val = calloc(n_alloc, (size_t)elem_size);
        t->col = (uint16_t *)calloc(n_alloc, sizeof(uint16_t));
        if (!t->val || !t->col)
            goto out;
    }
    spmv_parallel_for(rows, 1024, lanes_scatter, &c);
    if (c.bad)
        goto out;
    rc = 0;
out:
    free(c.cnt), free(c.estart), free(c.lane), free(c.wave), free(c.sbase), free(c.sslot), free(c.wlen);
    free(c.wnz), free(c.wrel);
    if (rc != 0)
        spmv_lanes_free(t);
    return rc;
}
