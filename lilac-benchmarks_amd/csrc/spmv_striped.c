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
    /* the top 16 bytes of the buffer are not x: its last element is kept at 0 and is what the filler
     * of an empty (row, stripe) segment multiplies with, so that a non-finite x entry cannot reach a
     * row that does not reference it (0 * Inf = NaN) */
    wmax -= 16 / elem_size;
    if (target_wgs < 1)
        target_wgs = 1;
    /*
     * Fewest stripes that fit LDS (fewest partial sums per row), all of the same width, then as many row ranges per
     * stripe as keep S * B within the target: every workgroup is resident at once, no tail round.
     * Narrower stripes that fill the last workgroup slots were measured and lose: on a class D row
     * block S = 125 (250 workgroups) ran 270 us against 250 us for S = 92 (184 workgroups), because
     * every extra stripe adds a row-end marker and a partial sum per row.
     */
    int s_min = cols > 0 ? (cols + wmax - 1) / wmax : 1;
    /* equal stripes: with full-width stripes and a remainder, the last stripe of NPB class C held
     * 1.7 % of the nonzeros but a tenth of the workgroups */
    const int unit = 16 / elem_size; /* stripes start on 16-byte boundaries of x */
    int best_w = cols > 0 ? ((cols + s_min - 1) / s_min + unit - 1) / unit * unit : wmax;
    if (best_w > wmax)
        best_w = wmax;
    if (best_w < unit)
        best_w = unit;
    s_min = cols > 0 ? (cols + best_w - 1) / best_w : 1;
    int best_s = s_min, best_b = target_wgs / s_min > 0 ? target_wgs / s_min : 1;
    if (best_b < 1)
        best_b = 1;
    /* no point in more waves than rows */
    while (best_b > 1 && (int64_t)best_b * wpb > (int64_t)rows)
        --best_b;
    t->W = best_w;
    t->S = best_s;
    t->B = best_b;
}

/* ---- the conversion, in passes that split over host threads --------------------------------- */

typedef struct build_ctx {
    const void *a;
    const int *rowstr, *colidx;
    int r0, rows, cols, elem, S, W, B, WPB, EPL, G;
    int cap;         /* elements of the kernel's LDS buffer: its last one is the fillers' zero */
    int64_t SUB;
    int32_t *cnt;    /* [S][rows] entries of row r in stripe s                               */
    int32_t *start;  /* [S][rows] position of the segment's first entry, relative to its stripe */
    int64_t *sbase;  /* [S+1]     first position of each stripe                               */
    int64_t *sdummy; /* [S]       fillers per stripe                                           */
    int32_t *wlen;   /* [nw]      wave run lengths (whole sub-windows), then absolute via wptr */
    spmv_striped_host *t;
    volatile int bad; /* a column outside [1, cols] was seen */
} build_ctx;

/* pass 1, over rows: count the entries per (stripe, row) */
static void pass_count(int64_t lo, int64_t hi, void *vc)
{
    build_ctx *c = (build_ctx *)vc;
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

/* pass 2, over stripes: cut the rows of a stripe into B*WPB ranges of about equal entry counts and
 * place every segment relative to the stripe's start (every wave streams whole sub-windows) */
static void pass_layout(int64_t lo, int64_t hi, void *vc)
{
    build_ctx *c = (build_ctx *)vc;
    const int rows = c->rows, parts = c->B * c->WPB;
    for (int64_t s = lo; s < hi; ++s) {
        const int32_t *cs = c->cnt + (size_t)s * rows;
        int32_t *st = c->start + (size_t)s * rows;
        int64_t stripe_entries = 0, dummies = 0;
        for (int r = 0; r < rows; ++r)
            stripe_entries += cs[r] > 0 ? cs[r] : 1;
        int r = 0;
        int64_t done = 0, total = 0;
        for (int p = 0; p < parts; ++p) {
            const int wi = (int)s * parts + p;
            const int64_t wave_begin = total;
            c->t->wrow[wi] = r;
            const int64_t target = stripe_entries * (p + 1) / parts;
            /* leave at least one row for every remaining part when possible */
            while (r < rows && (done < target || p == parts - 1)) {
                if (p < parts - 1 && rows - r <= parts - 1 - p)
                    break;
                const int n = cs[r] > 0 ? cs[r] : 1;
                dummies += cs[r] == 0;
                st[r] = (int32_t)total;
                total += n;
                done += n;
                ++r;
            }
            total = (total + c->SUB - 1) / c->SUB * c->SUB;
            c->wlen[wi] = (int32_t)(total - wave_begin);
        }
        c->sbase[s + 1] = total; /* turned into a running sum by the caller */
        c->sdummy[s] = dummies;
    }
}

/* pass 3, over rows: scatter in CSR order; a row's segments are touched by its own thread only */
static void pass_scatter(int64_t lo, int64_t hi, void *vc)
{
    build_ctx *c = (build_ctx *)vc;
    const int rows = c->rows, S = c->S, W = c->W, EPL = c->EPL, G = c->G;
    const int64_t SUB = c->SUB;
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
            const int64_t dst = c->sbase[s] + c->start[k] + fill[s]++;
            uint16_t cc = (uint16_t)(col - s * W);
            if (fill[s] == c->cnt[k])
                cc |= 0x8000u; /* last entry of this row in this stripe */
            tcol[dst] = cc;
            /* values: lane-interleaved in 16-byte units inside each sub-window, so that lane l
             * (entries l*EPL .. l*EPL+EPL-1) fetches them with dense 16-byte-per-lane loads */
            const int64_t blk = dst / SUB * SUB, rr = dst % SUB;
            const int64_t l = rr / EPL, i = rr % EPL;
            const int64_t vdst = blk + (i / G) * (64 * G) + l * G + (i % G);
            if (c->elem == 8)
                ((double *)c->t->val)[vdst] = ((const double *)c->a)[j];
            else
                ((float *)c->t->val)[vdst] = ((const float *)c->a)[j];
        }
    }
    free(fill);
}

/* pass 4, over stripes: the filler of an empty segment is a flag-only entry: value 0, and as its
 * column the last element of the LDS buffer, which the kernel keeps at 0 (spmv_striped_geometry) */
static void pass_fillers(int64_t lo, int64_t hi, void *vc)
{
    build_ctx *c = (build_ctx *)vc;
    for (int64_t s = lo; s < hi; ++s) {
        if (c->sdummy[s] == 0)
            continue;
        const int32_t *cs = c->cnt + (size_t)s * c->rows;
        const int32_t *st = c->start + (size_t)s * c->rows;
        for (int r = 0; r < c->rows; ++r)
            if (cs[r] == 0)
                c->t->col[c->sbase[s] + st[r]] = (uint16_t)(0x8000u | (unsigned)(c->cap - 1));
    }
}

/*
 * rowstr / colidx / a: the caller's 1-based CSR; rows [r0,r1) are converted.
 * Returns 0, -1 (out of memory), -2 (not representable / not worth it).
 * The result does not depend on the number of threads (SPMV_HIP_BUILD_THREADS).
 */
int spmv_striped_build(const void *a, const int *rowstr, const int *colidx, int r0, int r1, int cols,
                       int elem_size, int xbytes, int wpb, int target_wgs, int epl, spmv_striped_host *t)
{
    const int rows = r1 - r0;
    spmv_striped_geometry(rows, cols, elem_size, xbytes, wpb, target_wgs, epl, t);
    const int S = t->S;
    if (rows <= 0 || (int64_t)S * rows > ((int64_t)1 << 28))
        return -2;
    const int64_t nnz = (int64_t)rowstr[r1] - rowstr[r0];
    const int nw = S * t->B * t->WPB;

    build_ctx c;
    memset(&c, 0, sizeof c);
    c.a = a, c.rowstr = rowstr, c.colidx = colidx;
    c.r0 = r0, c.rows = rows, c.cols = cols, c.elem = elem_size;
    c.S = S, c.W = t->W, c.B = t->B, c.WPB = t->WPB, c.EPL = t->EPL;
    c.G = 16 / elem_size;        /* values per 16-byte unit */
    c.cap = xbytes / elem_size > 32768 ? 32768 : xbytes / elem_size;
    c.SUB = 64 * (int64_t)t->EPL; /* entries per sub-window */
    c.t = t;
    int rc = -1;
    c.cnt = (int32_t *)calloc((size_t)S * rows, sizeof(int32_t));
    c.start = (int32_t *)malloc((size_t)S * rows * sizeof(int32_t));
    c.sbase = (int64_t *)calloc((size_t)S + 1, sizeof(int64_t));
    c.sdummy = (int64_t *)calloc((size_t)S, sizeof(int64_t));
    c.wlen = (int32_t *)calloc((size_t)nw + 1, sizeof(int32_t));
    t->n_waves = nw;
    t->wptr = (int32_t *)calloc((size_t)nw + 1, sizeof(int32_t));
    t->wrow = (int32_t *)calloc((size_t)nw + 1, sizeof(int32_t));
    if (!c.cnt || !c.start || !c.sbase || !c.sdummy || !c.wlen || !t->wptr || !t->wrow)
        goto out;

    spmv_parallel_for(rows, 4096, pass_count, &c);
    if (c.bad) {
        rc = -2;
        goto out;
    }
    spmv_parallel_for(S, 1, pass_layout, &c);

    int64_t total = 0, dummies = 0;
    for (int s = 0; s < S; ++s) {
        const int64_t len = c.sbase[s + 1];
        c.sbase[s] = total;
        total += len;
        dummies += c.sdummy[s];
    }
    c.sbase[S] = total;
    if (total > 0x7fffffff - 65536) {
        rc = -2;
        goto out;
    }
    {
        int64_t pos = 0;
        for (int wi = 0; wi < nw; ++wi) {
            t->wptr[wi] = (int32_t)pos;
            pos += c.wlen[wi];
        }
        t->wptr[nw] = (int32_t)pos; /* == total */
        t->wrow[nw] = rows;
    }
    t->n_packed = total;
    t->n_dummies = dummies;
    if (dummies * 8 > nnz) { /* more than 1/8 of the stream would be filler: the matrix is too
                                sparse per stripe for this plan */
        rc = -2;
        goto out;
    }

    {
        const size_t n_alloc = (size_t)total + SPMV_TILE_PAD;
        t->val = calloc(n_alloc, (size_t)elem_size);
        t->col = (uint16_t *)calloc(n_alloc, sizeof(uint16_t));
        if (!t->val || !t->col)
            goto out;
    }
    spmv_parallel_for(rows, 1024, pass_scatter, &c);
    if (c.bad)
        goto out;
    spmv_parallel_for(S, 1, pass_fillers, &c);
    rc = 0;
out:
    free(c.cnt), free(c.start), free(c.sbase), free(c.sdummy), free(c.wlen);
    if (rc != 0)
        spmv_striped_free(t);
    return rc;
}
