/*
 * csrc/spmv_striped.c -- host-side conversion of a CSR row block into the STRIPED layout used
 * by k_striped (plain C, no GPU needed).
 *
 * STRIPED is the 2-D decomposition of y = A*x.  The columns are cut into S stripes of W columns
 * (W * sizeof(T) <= 128 KiB of x: one stripe fits the LDS of a CU) and the rows into NB row blocks
 * of WPB wave ranges each (a wave range = at most `wave_rows` consecutive rows, all ranges of about
 * equal nonzero counts, the SAME cuts for every stripe).  The stripes are dealt to NG groups of
 * consecutive stripes, and a workgroup = (stripe group g, row block rb).  It walks the stripes of its
 * group one after the other: copy the stripe of x into LDS, then each of its WPB waves streams the
 * entries of ITS rows that fall into that stripe -- the per-nonzero gather of x is an LDS access --
 * and adds each row's sum for the stripe into a per-row accumulator that lives in the wave's
 * registers across the whole group (lane l, register k <-> row 64 k + l of the wave range).  Only
 * when the group is done does a row sum leave the CU: one partial sum per (stripe GROUP, row), or y
 * itself when NG = 1.
 *
 * Why groups: with one stripe per workgroup (round 1) a row cost one partial sum per STRIPE; on NPB
 * class D (1.5 M columns, 92 stripes, 5 nonzeros per row and stripe) writing and re-reading them was
 * 24 % of all bytes moved.  With groups, the partial sums shrink by the group length; k_group_sum adds
 * the NG partials of a row in group order (a second, small launch: see the note there on why not the
 * last workgroup to arrive).
 *
 * Layout (offsets in entries):
 *
 *   tile (g, rb, w, si)   the entries of wave range (rb, w) in stripe s = s0(g) + si, row after row,
 *                         each row's entries in their original CSR order; a whole number of
 *                         sub-windows of 64*EPL entries (pads: value 0, the zero column, no flag);
 *                         a wave's tiles follow each other in memory (one linear stream per wave per
 *                         workgroup), waves of a workgroup and workgroups (g-major) follow each other;
 *   tptr[]                int32 [(NG*NB*WPB) * (GM + 1)]: tile starts of wave (g, rb, w), GM =
 *                         ceil(S / NG) stripes per group at most, entry GM (and those past the
 *                         group's last stripe) = end of the wave's stream;
 *   rrow[]                int32 [NB*WPB + 1]: first row of every wave range;
 *   val[]                 T, packed; inside each sub-window lane-interleaved in 16-byte units: entry
 *                         e = l*EPL + i sits at (i/G)*(64*G) + l*G + i%G, G = 16/sizeof(T), so that
 *                         lane l, which owns entries l*EPL.., reads them with dense 16-byte loads;
 *   col[]                 uint16: bits 0..14 = column - 1 - s*W, bit 15 = "last entry of its row in
 *                         this stripe".  Every row owns exactly one flagged entry per stripe -- a row
 *                         with no nonzero in the stripe gets a filler (value 0, the zero column, flag
 *                         set) -- so a wave recovers row numbers by counting flags and reads no row
 *                         pointers at all.  The zero column is the last element of the kernel's LDS
 *                         buffer, which is kept at 0: a non-finite x entry cannot reach a row that
 *                         does not reference it.
 *
 * Summation order: inside a (row, stripe) segment a fixed tree (EPL-entry lane sums, then a segmented
 * wave scan), across the stripes of a group ascending, across groups ascending.  Deterministic, but
 * not the reference's left-to-right order: results agree with native-impl.c within the reordering
 * bound, not bit for bit (TILED and ORDERED are the bit-exact plans).
 */
#include "spmv_internal.h"

#include <stdlib.h>
#include <string.h>

void spmv_striped_free(spmv_striped_host *t)
{
    free(t->val);
    free(t->col);
    free(t->tptr);
    free(t->rrow);
    memset(t, 0, sizeof *t);
}

void spmv_striped_default_opts(spmv_striped_opts *o)
{
    o->xbytes = SPMV_STRIPE_XBYTES;
    o->wpb = SPMV_STRIPE_WPB;
    o->target_wgs = SPMV_STRIPE_TARGET_WGS;
    o->wave_rows = SPMV_STRIPE_WAVE_ROWS;
    o->groups = 0;
}

void spmv_striped_geometry(int rows, int cols, int elem_size, const spmv_striped_opts *o, spmv_striped_host *t)
{
    memset(t, 0, sizeof *t);
    t->elem = elem_size;
    t->EPL = SPMV_STRIPE_EPL;
    t->WPB = o->wpb;
    int wmax = o->xbytes / elem_size; /* columns that fit the LDS stripe buffer */
    if (wmax > 32768)
        wmax = 32768; /* 15 column bits */
    t->cap = wmax;
    /* the top 16 bytes of the buffer are not x: its last element is kept at 0 and is what fillers and pads
     * multiply with */
    wmax -= 16 / elem_size;
    /* fewest stripes that fit LDS, all of the same width (with full-width stripes and a remainder, the last
     * stripe of NPB class C held 1.7 % of the nonzeros but a tenth of the workgroups) */
    int s_min = cols > 0 ? (cols + wmax - 1) / wmax : 1;
    const int unit = 16 / elem_size; /* stripes start on 16-byte boundaries of x */
    int w = cols > 0 ? ((cols + s_min - 1) / s_min + unit - 1) / unit * unit : wmax;
    if (w > wmax)
        w = wmax;
    if (w < unit)
        w = unit;
    t->W = w;
    t->S = cols > 0 ? (cols + w - 1) / w : 1;
    /*
     * Row blocks and stripe groups.  A wave holds at most wave_rows row accumulators, which sets the
     * least number of row blocks; within that, as many stripe groups as keep NB * NG within the target
     * (every workgroup resident at once: one 160-KiB workgroup per CU), because fewer stripes per
     * workgroup mean fewer re-stagings of x and longer uninterrupted streams, and the partial sums
     * that more groups cost are few: class C 10 groups x 25 row blocks (12 MB of partials beside a
     * 373 MB stream), a class D row block of 187 500 rows 21 x 12 (4-5 stripes per group), all of
     * class D 2 x 128.
     */
    const int target = o->target_wgs < 1 ? 1 : o->target_wgs;
    const int64_t per_block = (int64_t)o->wpb * o->wave_rows;
    int nb_min = rows > 0 ? (int)((rows + per_block - 1) / per_block) : 1;
    int ng = o->groups > 0 ? o->groups : target / nb_min;
    if (ng < 1)
        ng = 1;
    if (ng > t->S)
        ng = t->S;
    int nb = target / ng;
    if (nb < nb_min)
        nb = nb_min;
    while (nb > nb_min && (int64_t)nb * o->wpb > (int64_t)rows) /* no point in more waves than rows */
        --nb;
    t->NB = nb;
    t->NG = ng;
    t->GM = (t->S + ng - 1) / ng;
    t->n_ranges = nb * o->wpb;
}

/* exact c / W for 0 <= c < 2^31 without a hardware divide per nonzero */
typedef struct fastdiv {
    uint64_t m;
    uint32_t w;
} fastdiv;
static fastdiv fastdiv_make(int w)
{
    fastdiv d;
    d.w = (uint32_t)w;
    d.m = ((uint64_t)1 << 40) / (uint64_t)w + 1;
    return d;
}
static inline uint32_t fastdiv_do(fastdiv d, uint32_t c)
{
    uint32_t q = (uint32_t)(((uint64_t)c * d.m) >> 40); /* exact or one too large */
    if ((uint64_t)q * d.w > c)
        --q;
    return q;
}

/* ---- the conversion, in passes that split over host threads --------------------------------- */

typedef struct build_ctx {
    const void *a;
    const int *rowstr, *colidx;
    int r0, rows, cols, elem, S, W, NB, NG, GM, WPB, EPL, G, cap;
    int64_t SUB;
    fastdiv div;
    int64_t *tsize;   /* [n_ranges][S] padded tile sizes */
    int64_t *rfill;   /* [n_ranges] fillers */
    int *sg, *ssi;    /* [S] group of a stripe and its index inside the group */
    spmv_striped_host *t;
    volatile int bad; /* 1: a column outside [1, cols]; 2: out of memory */
} build_ctx;

/* pass 1, over wave ranges: entries (fillers included) per (range, stripe) */
static void pass_count(int64_t lo, int64_t hi, void *vc)
{
    build_ctx *c = (build_ctx *)vc;
    const int S = c->S;
    int32_t *stamp = (int32_t *)malloc((size_t)S * sizeof(int32_t));
    if (!stamp) {
        c->bad = 2;
        return;
    }
    for (int64_t u = lo; u < hi; ++u) {
        int64_t *cnt = c->tsize + (size_t)u * S;
        memset(cnt, 0, (size_t)S * sizeof(int64_t));
        for (int s = 0; s < S; ++s)
            stamp[s] = -1;
        int64_t fillers = 0;
        for (int r = c->t->rrow[u]; r < c->t->rrow[u + 1]; ++r) {
            int distinct = 0;
            for (int j = c->rowstr[c->r0 + r] - 1; j < c->rowstr[c->r0 + r + 1] - 1; ++j) {
                const int col = c->colidx[j] - 1;
                if (col < 0 || col >= c->cols) {
                    c->bad = 1;
                    free(stamp);
                    return;
                }
                const uint32_t s = fastdiv_do(c->div, (uint32_t)col);
                cnt[s]++;
                if (stamp[s] != r)
                    stamp[s] = r, ++distinct;
            }
            if (distinct < S) {
                for (int s = 0; s < S; ++s)
                    cnt[s] += stamp[s] != r;
                fillers += S - distinct;
            }
        }
        c->rfill[u] = fillers;
        for (int s = 0; s < S; ++s)
            cnt[s] = (cnt[s] + c->SUB - 1) / c->SUB * c->SUB;
    }
    free(stamp);
}

static inline int64_t val_slot(const build_ctx *c, int64_t p)
{
    /* values: lane-interleaved in 16-byte units inside each sub-window, so that lane l (entries
     * l*EPL .. l*EPL+EPL-1) fetches them with dense 16-byte-per-lane loads */
    const int64_t blk = p / c->SUB * c->SUB, rr = p % c->SUB;
    const int64_t l = rr / c->EPL, i = rr % c->EPL;
    return blk + (i / c->G) * (64 * c->G) + l * c->G + (i % c->G);
}

/* pass 2, over wave ranges: scatter in CSR order; a range's tiles are touched by its own thread only */
static void pass_scatter(int64_t lo, int64_t hi, void *vc)
{
    build_ctx *c = (build_ctx *)vc;
    const int S = c->S, W = c->W;
    int32_t *stamp = (int32_t *)malloc((size_t)S * sizeof(int32_t));
    int64_t *pos = (int64_t *)malloc((size_t)S * sizeof(int64_t));
    int64_t *last = (int64_t *)malloc((size_t)S * sizeof(int64_t));
    if (!stamp || !pos || !last) {
        c->bad = 2;
        free(stamp), free(pos), free(last);
        return;
    }
    uint16_t *tcol = c->t->col;
    const uint16_t zero_col = (uint16_t)(c->cap - 1);
    const int nr = c->t->n_ranges;
    for (int64_t u = lo; u < hi; ++u) {
        for (int s = 0; s < S; ++s) {
            pos[s] = c->t->tptr[((size_t)c->sg[s] * nr + (size_t)u) * ((size_t)c->GM + 1) + c->ssi[s]];
            stamp[s] = -1;
        }
        for (int r = c->t->rrow[u]; r < c->t->rrow[u + 1]; ++r) {
            int distinct = 0;
            for (int j = c->rowstr[c->r0 + r] - 1; j < c->rowstr[c->r0 + r + 1] - 1; ++j) {
                const int col = c->colidx[j] - 1;
                const uint32_t s = fastdiv_do(c->div, (uint32_t)col);
                const int64_t p = pos[s]++;
                tcol[p] = (uint16_t)(col - (int)s * W);
                const int64_t v = val_slot(c, p);
                if (c->elem == 8)
                    ((double *)c->t->val)[v] = ((const double *)c->a)[j];
                else
                    ((float *)c->t->val)[v] = ((const float *)c->a)[j];
                last[s] = p;
                if (stamp[s] != r)
                    stamp[s] = r, ++distinct;
            }
            if (distinct == S) {
                for (int s = 0; s < S; ++s)
                    tcol[last[s]] |= 0x8000u; /* last entry of this row in this stripe */
            } else {
                for (int s = 0; s < S; ++s)
                    if (stamp[s] == r)
                        tcol[last[s]] |= 0x8000u;
                    else /* the filler of an empty segment: flag only, value 0 (calloc), times the zero column */
                        tcol[pos[s]++] = (uint16_t)(0x8000u | zero_col);
            }
        }
        /* the rest of every tile is padding: value 0 (calloc) times the zero column, no flag */
        for (int s = 0; s < S; ++s) {
            const int64_t end = c->t->tptr[((size_t)c->sg[s] * nr + (size_t)u) * ((size_t)c->GM + 1) + c->ssi[s] + 1];
            for (int64_t p = pos[s]; p < end; ++p)
                tcol[p] = zero_col;
        }
    }
    free(stamp), free(pos), free(last);
}

/*
 * The wave ranges: NB*WPB runs of consecutive rows of about equal weight (nonzeros + 1), none longer
 * than wave_rows rows, every one non-empty while rows last.
 */
static void cut_rows(const int *rowstr, int r0, int rows, int n_ranges, int wave_rows, int32_t *rrow)
{
    int64_t remaining = (int64_t)rowstr[r0 + rows] - rowstr[r0] + rows;
    int r = 0;
    for (int i = 0; i < n_ranges; ++i) {
        rrow[i] = r;
        const int left = n_ranges - i;
        const int64_t target = (remaining + left - 1) / left;
        int64_t acc = 0;
        while (r < rows) {
            const int64_t w = (int64_t)rowstr[r0 + r + 1] - rowstr[r0 + r] + 1;
            const int taken = r - rrow[i];
            /* the ranges after this one hold (left - 1) * wave_rows rows at most */
            const int must = (int64_t)(rows - r) > (int64_t)(left - 1) * wave_rows;
            if (!must) {
                if (taken >= wave_rows)
                    break;
                if (taken >= 1 && rows - r <= left - 1)
                    break; /* leave a row for every range that follows */
                if (taken >= 1 && acc + w / 2 >= target)
                    break;
            }
            acc += w;
            ++r;
        }
        remaining -= acc;
    }
    rrow[n_ranges] = rows;
}

/*
 * rowstr / colidx / a: the caller's 1-based CSR; rows [r0,r1) are converted.
 * Returns 0, -1 (out of memory), -2 (not representable / not worth it).
 * The result does not depend on the number of threads (SPMV_HIP_BUILD_THREADS).
 */
int spmv_striped_build(const void *a, const int *rowstr, const int *colidx, int r0, int r1, int cols,
                       int elem_size, const spmv_striped_opts *opts, spmv_striped_host *t)
{
    const int rows = r1 - r0;
    spmv_striped_opts dflt;
    if (!opts) {
        spmv_striped_default_opts(&dflt);
        opts = &dflt;
    }
    spmv_striped_geometry(rows, cols, elem_size, opts, t);
    const int S = t->S;
    if (rows <= 0 || opts->wave_rows < 64 || opts->wave_rows % 64 || S > 32768 /* fastdiv_do */ ||
        (int64_t)t->n_ranges * S > ((int64_t)1 << 26))
        return -2;
    const int64_t nnz = (int64_t)rowstr[r1] - rowstr[r0];
    const int nr = t->n_ranges;

    build_ctx c;
    memset(&c, 0, sizeof c);
    c.a = a, c.rowstr = rowstr, c.colidx = colidx;
    c.r0 = r0, c.rows = rows, c.cols = cols, c.elem = elem_size;
    c.S = S, c.W = t->W, c.NB = t->NB, c.NG = t->NG, c.GM = t->GM, c.WPB = t->WPB, c.EPL = t->EPL;
    c.G = 16 / elem_size; /* values per 16-byte unit */
    c.cap = t->cap;
    c.SUB = 64 * (int64_t)t->EPL; /* entries per sub-window */
    c.div = fastdiv_make(t->W);
    c.t = t;
    int rc = -1;
    c.tsize = (int64_t *)malloc((size_t)nr * S * sizeof(int64_t));
    c.rfill = (int64_t *)calloc((size_t)nr, sizeof(int64_t));
    c.sg = (int *)malloc((size_t)S * sizeof(int));
    c.ssi = (int *)malloc((size_t)S * sizeof(int));
    t->rrow = (int32_t *)calloc((size_t)nr + 1, sizeof(int32_t));
    const size_t ntp = (size_t)t->NG * nr * ((size_t)t->GM + 1);
    t->tptr = (int32_t *)calloc(ntp, sizeof(int32_t));
    if (!c.tsize || !c.rfill || !c.sg || !c.ssi || !t->rrow || !t->tptr)
        goto out;
    for (int g = 0; g < t->NG; ++g)
        for (int s = (int)((int64_t)g * S / t->NG); s < (int)((int64_t)(g + 1) * S / t->NG); ++s)
            c.sg[s] = g, c.ssi[s] = s - (int)((int64_t)g * S / t->NG);

    cut_rows(rowstr, r0, rows, nr, opts->wave_rows, t->rrow);
    spmv_parallel_for(nr, 1, pass_count, &c);
    if (c.bad) {
        rc = c.bad == 1 ? -2 : -1;
        goto out;
    }

    /* tile starts in stream order: stripe group, row block, wave, stripe of the group */
    {
        int64_t total = 0, fillers = 0, longest = 0, ntiles = 0;
        for (int u = 0; u < nr; ++u)
            fillers += c.rfill[u];
        for (int g = 0; g < t->NG; ++g) {
            const int s0 = (int)((int64_t)g * S / t->NG), s1 = (int)((int64_t)(g + 1) * S / t->NG);
            for (int u = 0; u < nr; ++u) { /* u = rb * WPB + w */
                int32_t *tp = t->tptr + ((size_t)g * nr + u) * ((size_t)t->GM + 1);
                for (int si = 0; si <= t->GM; ++si) {
                    if (total > 0x7fffffff - 65536) {
                        rc = -2;
                        goto out;
                    }
                    tp[si] = (int32_t)total;
                    if (s0 + si < s1) {
                        const int64_t len = c.tsize[(size_t)u * S + s0 + si];
                        total += len;
                        longest = len > longest ? len : longest;
                        ++ntiles;
                    }
                }
            }
        }
        t->n_packed = total;
        t->n_dummies = fillers;
        t->max_tile = longest;
        if (fillers * 8 > nnz) { /* more than 1/8 of the stream would be filler: the matrix is too sparse per
                                    stripe for this plan */
            rc = -2;
            goto out;
        }
        /* a (row, stripe) segment is never split between waves: one dense row makes its tile many times longer
         * than the rest, and every wave of its workgroup waits for it at the stripe's end */
        t->skewed = longest * ntiles > 2 * total + c.SUB * ntiles;
    }

    {
        const size_t n_alloc = (size_t)t->n_packed + SPMV_TILE_PAD;
        t->val = calloc(n_alloc, (size_t)elem_size);
        t->col = (uint16_t *)malloc(n_alloc * sizeof(uint16_t));
        if (!t->val || !t->col)
            goto out;
        for (size_t i = (size_t)t->n_packed; i < n_alloc; ++i) /* the look-ahead past the last tile: pads too */
            t->col[i] = (uint16_t)(c.cap - 1);
    }
    spmv_parallel_for(nr, 1, pass_scatter, &c);
    if (c.bad)
        goto out;
    rc = 0;
out:
    free(c.tsize), free(c.rfill), free(c.sg), free(c.ssi);
    if (rc != 0)
        spmv_striped_free(t);
    return rc;
}
