/*
 * csrc/spmv_striped.c -- host-side conversion of a CSR row block into the STRIPED layout used
 * by k_striped (plain C, no GPU needed).
 *
 * STRIPED is the 2-D decomposition of y = A*x.  The columns are cut into S stripes of W columns
 * (W * sizeof(T) <= 128 KiB of x: one stripe fits the LDS of a CU), and the stripes are dealt to NG
 * groups of consecutive stripes.  Every group cuts the rows into its own row blocks of WPB wave ranges
 * each (a wave range = at most `wave_rows` consecutive rows; all ranges of a group hold the same number
 * of entries OF THAT GROUP's stripes, and a group gets row blocks in proportion to its entries), and a
 * workgroup = (stripe group g, row block rb of g).  It walks the stripes of its
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
 *   tile (q, w, si)       workgroup q = (g, rb), wave w, stripe s = s0(g) + si: the entries of the wave's
 *                         rows in that stripe, row after row, each row's entries in their original
 *                         CSR order; a whole number of sub-windows of 64*EPL entries (pads: value 0,
 *                         the zero column, no flag); a wave's tiles follow each other in memory (one
 *                         linear stream per wave), waves and workgroups (g-major) follow each other;
 *   wgd[]                 int32 [n_wg * 4]: first stripe, end stripe and group of workgroup q;
 *   tptr[]                int32 [(n_wg*WPB) * (GM + 1)]: tile starts of wave u = q*WPB + w, GM =
 *                         ceil(S / NG) stripes per group at most, entry GM (and those past the
 *                         group's last stripe) = end of the wave's stream;
 *   rrow[]                int32 [n_wg*WPB * 2]: first and end row of wave u;
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
    free(t->wgd);
    memset(t, 0, sizeof *t);
}

void spmv_striped_default_opts(spmv_striped_opts *o)
{
    o->xbytes = SPMV_STRIPE_XBYTES;
    o->wpb = SPMV_STRIPE_WPB;
    o->target_wgs = SPMV_STRIPE_TARGET_WGS;
    o->wave_rows = SPMV_STRIPE_WAVE_ROWS;
    o->groups = 0;
    o->stripes = 0;
    o->xcd_bias = 0;
}

int spmv_striped_choose_shape(const spmv_row_stats *st, int elem_size, int cols)
{
    (void)st, (void)elem_size, (void)cols;
    return 1;
}

/* stripes and stripe groups; the row blocks per group are decided by the builder, from the entries */
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
    const int unit = 16 / elem_size; /* stripes start on 16-byte boundaries of x */
    int s_min = cols > 0 ? (cols + wmax - 1) / wmax : 1;
    if (o->stripes > s_min) /* more, narrower stripes than the buffer demands: so that they divide into the groups */
        s_min = o->stripes;
    /*
     * Stripe groups.  A wave holds at most wave_rows row accumulators, which sets the least number of row
     * blocks; within that, as many stripe groups as keep the workgroups within the target (every workgroup
     * resident at once: one 160-KiB workgroup per CU), because fewer stripes per workgroup mean fewer
     * re-stagings of x and longer uninterrupted streams, and the partial sums that more groups cost are
     * few (class C: 12 MB of partials beside a 373 MB stream).  This is the upper bound; the builder lowers it
     * where the groups' row blocks would not fit the target (choose_groups).
     */
    const int target = o->target_wgs < 1 ? 1 : o->target_wgs;
    const int64_t per_block = (int64_t)o->wpb * o->wave_rows;
    const int nb_min = rows > 0 ? (int)((rows + per_block - 1) / per_block) : 1;
    int ng = o->groups > 0 ? o->groups : target / nb_min;
    if (ng < 1)
        ng = 1;
    if (ng > s_min)
        ng = s_min;
    /* all stripes of the same width (with full-width stripes and a remainder, the last stripe of NPB class C
     * held 1.7 % of the nonzeros but a tenth of the workgroups) */
    int w = cols > 0 ? ((cols + s_min - 1) / s_min + unit - 1) / unit * unit : wmax;
    if (w > wmax)
        w = wmax;
    if (w < unit)
        w = unit;
    t->W = w;
    t->S = cols > 0 ? (cols + w - 1) / w : 1;
    if (ng > t->S)
        ng = t->S;
    t->NG = ng;
    t->GM = (t->S + ng - 1) / ng;
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
    int r0, rows, cols, elem, S, W, NG, GM, WPB, EPL, G, cap;
    int64_t SUB;
    fastdiv div;
    int *sg;          /* [S] group of a stripe                                                  */
    int *gs0;         /* [NG + 1] first stripe of a group                                       */
    uint32_t *wgt;    /* [NG][rows] entries (fillers included) of a row in a group's stripes    */
    int64_t *tsize;   /* [n_ranges][GM] padded tile sizes                                       */
    int64_t *rfill;   /* [n_ranges] fillers                                                     */
    spmv_striped_host *t;
    int unsorted;          /* some row's columns decrease: a group's entries are not one run of the row */
    int bad;               /* 1: a column outside [1, cols]; 2: out of memory (set by any thread, read after the pass) */
} build_ctx;

/* pass 1, over rows: entries + fillers of every row in every stripe group; are the rows sorted? */
static void pass_weigh(int64_t lo, int64_t hi, void *vc)
{
    build_ctx *c = (build_ctx *)vc;
    const int S = c->S, NG = c->NG;
    int64_t *stamp = (int64_t *)malloc((size_t)S * sizeof(int64_t));
    uint32_t *cnt = (uint32_t *)malloc((size_t)NG * 2 * sizeof(uint32_t));
    if (!stamp || !cnt) {
        __atomic_store_n(&c->bad, 2, __ATOMIC_RELAXED);
        free(stamp), free(cnt);
        return;
    }
    uint32_t *touched = cnt + NG;
    for (int s = 0; s < S; ++s)
        stamp[s] = -1;
    int unsorted = 0;
    for (int64_t r = lo; r < hi; ++r) {
        memset(cnt, 0, (size_t)NG * 2 * sizeof(uint32_t));
        int prev = 0;
        for (int j = c->rowstr[c->r0 + r] - 1; j < c->rowstr[c->r0 + r + 1] - 1; ++j) {
            const int col = c->colidx[j] - 1;
            if (col < 0 || col >= c->cols) {
                __atomic_store_n(&c->bad, 1, __ATOMIC_RELAXED);
                free(stamp), free(cnt);
                return;
            }
            unsorted |= col < prev;
            prev = col;
            const uint32_t s = fastdiv_do(c->div, (uint32_t)col);
            const int g = c->sg[s];
            cnt[g]++;
            if (stamp[s] != r)
                stamp[s] = r, touched[g]++;
        }
        for (int g = 0; g < NG; ++g)
            c->wgt[(size_t)g * c->rows + r] = cnt[g] + (uint32_t)(c->gs0[g + 1] - c->gs0[g]) - touched[g];
    }
    if (unsorted) /* several threads may find this out: the same value from all of them */
        __atomic_store_n(&c->unsorted, 1, __ATOMIC_RELAXED);
    free(stamp), free(cnt);
}

/* the entries of row r (local) that fall into stripes [s0, s1): a run [*jb, *je) of the row when its columns are
 * sorted; otherwise the whole row, to be filtered */
static inline void group_run(const build_ctx *c, int r, int s0, int s1, int *jb, int *je)
{
    int b = c->rowstr[c->r0 + r] - 1, e = c->rowstr[c->r0 + r + 1] - 1;
    if (!c->unsorted) {
        const int64_t clo = (int64_t)s0 * c->W + 1, chi = (int64_t)s1 * c->W + 1; /* 1-based bounds */
        int lo = b, hi = e;
        while (lo < hi) { /* first entry with colidx >= clo */
            const int mid = lo + (hi - lo) / 2;
            if (c->colidx[mid] < clo)
                lo = mid + 1;
            else
                hi = mid;
        }
        b = lo;
        hi = e;
        while (lo < hi) { /* first entry with colidx >= chi */
            const int mid = lo + (hi - lo) / 2;
            if (c->colidx[mid] < chi)
                lo = mid + 1;
            else
                hi = mid;
        }
        e = lo;
    }
    *jb = b, *je = e;
}

/* pass 2, over wave ranges: entries (fillers included) per (range, stripe of its group) */
static void pass_count(int64_t lo, int64_t hi, void *vc)
{
    build_ctx *c = (build_ctx *)vc;
    const int GM = c->GM;
    int64_t *seen = (int64_t *)malloc((size_t)GM * sizeof(int64_t));
    if (!seen) {
        __atomic_store_n(&c->bad, 2, __ATOMIC_RELAXED);
        return;
    }
    for (int64_t u = lo; u < hi; ++u) {
        const int32_t *d = c->t->wgd + (size_t)(u / c->WPB) * 4;
        const int s0 = d[0], s1 = d[1], ns = s1 - s0;
        int64_t *cnt = c->tsize + (size_t)u * GM;
        memset(cnt, 0, (size_t)GM * sizeof(int64_t));
        for (int k = 0; k < GM; ++k)
            seen[k] = -1;
        int64_t fillers = 0;
        for (int r = c->t->rrow[2 * u]; r < c->t->rrow[2 * u + 1]; ++r) {
            int jb, je, distinct = 0;
            group_run(c, r, s0, s1, &jb, &je);
            for (int j = jb; j < je; ++j) {
                const int s = (int)fastdiv_do(c->div, (uint32_t)(c->colidx[j] - 1));
                if (s < s0 || s >= s1)
                    continue; /* unsorted rows: not this group's entry */
                cnt[s - s0]++;
                if (seen[s - s0] != r)
                    seen[s - s0] = r, ++distinct;
            }
            if (distinct < ns) {
                for (int k = 0; k < ns; ++k)
                    cnt[k] += seen[k] != r;
                fillers += ns - distinct;
            }
        }
        c->rfill[u] = fillers;
        for (int k = 0; k < GM; ++k)
            cnt[k] = (cnt[k] + c->SUB - 1) / c->SUB * c->SUB;
    }
    free(seen);
}

static inline int64_t val_slot(const build_ctx *c, int64_t p)
{
    /* values: lane-interleaved in 16-byte units inside each sub-window, so that lane l (entries
     * l*EPL .. l*EPL+EPL-1) fetches them with dense 16-byte-per-lane loads */
    const int64_t blk = p / c->SUB * c->SUB, rr = p % c->SUB;
    const int64_t l = rr / c->EPL, i = rr % c->EPL;
    return blk + (i / c->G) * (64 * c->G) + l * c->G + (i % c->G);
}

/* pass 3, over wave ranges: scatter in CSR order; a range's tiles are touched by its own thread only */
static void pass_scatter(int64_t lo, int64_t hi, void *vc)
{
    build_ctx *c = (build_ctx *)vc;
    const int GM = c->GM, W = c->W;
    int64_t *seen = (int64_t *)malloc((size_t)GM * 3 * sizeof(int64_t));
    if (!seen) {
        __atomic_store_n(&c->bad, 2, __ATOMIC_RELAXED);
        return;
    }
    int64_t *pos = seen + GM, *last = seen + 2 * GM;
    uint16_t *tcol = c->t->col;
    const uint16_t zero_col = (uint16_t)(c->cap - 1);
    for (int64_t u = lo; u < hi; ++u) {
        const int32_t *d = c->t->wgd + (size_t)(u / c->WPB) * 4;
        const int s0 = d[0], s1 = d[1], ns = s1 - s0;
        const int32_t *tp = c->t->tptr + (size_t)u * ((size_t)GM + 1);
        for (int k = 0; k < ns; ++k)
            pos[k] = tp[k], seen[k] = -1;
        for (int r = c->t->rrow[2 * u]; r < c->t->rrow[2 * u + 1]; ++r) {
            int jb, je, distinct = 0;
            group_run(c, r, s0, s1, &jb, &je);
            for (int j = jb; j < je; ++j) {
                const int col = c->colidx[j] - 1;
                const int s = (int)fastdiv_do(c->div, (uint32_t)col);
                if (s < s0 || s >= s1)
                    continue;
                const int k = s - s0;
                const int64_t p = pos[k]++;
                tcol[p] = (uint16_t)(col - s * W);
                const int64_t v = val_slot(c, p);
                if (c->elem == 8)
                    ((double *)c->t->val)[v] = ((const double *)c->a)[j];
                else
                    ((float *)c->t->val)[v] = ((const float *)c->a)[j];
                last[k] = p;
                if (seen[k] != r)
                    seen[k] = r, ++distinct;
            }
            if (distinct == ns) {
                for (int k = 0; k < ns; ++k)
                    tcol[last[k]] |= 0x8000u; /* last entry of this row in this stripe */
            } else {
                for (int k = 0; k < ns; ++k)
                    if (seen[k] == r)
                        tcol[last[k]] |= 0x8000u;
                    else /* the filler of an empty segment: flag only, value 0 (calloc), times the zero column */
                        tcol[pos[k]++] = (uint16_t)(0x8000u | zero_col);
            }
        }
        /* the rest of every tile is padding: value 0 (calloc) times the zero column, no flag */
        for (int k = 0; k < ns; ++k)
            for (int64_t p = pos[k]; p < tp[k + 1]; ++p)
                tcol[p] = zero_col;
    }
    free(seen);
}

/*
 * The wave ranges of one group: n runs of consecutive rows of about equal weight (the group's entries per row),
 * none longer than wave_rows rows, every one non-empty while rows last.  rr[2 i], rr[2 i + 1] = first / end row.
 */
/* share[i] (per mille, NULL = all equal): the part of the weight range i should get relative to the others -- the ranges
 * of workgroups that run on a slower XCD get a little less (spmv_striped_cut_all) */
static void cut_rows(const uint32_t *wgt, int rows, int n, int wave_rows, const int *share, int32_t *rr)
{
    int64_t remaining = 0, share_left = 0;
    for (int r = 0; r < rows; ++r)
        remaining += wgt[r];
    for (int i = 0; i < n; ++i)
        share_left += share ? share[i] : 1000;
    int r = 0;
    for (int i = 0; i < n; ++i) {
        const int first = r, left = n - i;
        const int64_t mine = share ? share[i] : 1000;
        const int64_t target = share_left > 0 ? (remaining * mine + share_left - 1) / share_left : 0;
        share_left -= mine;
        int64_t acc = 0;
        while (r < rows) {
            const int64_t w = wgt[r];
            const int taken = r - first;
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
        rr[2 * i] = first, rr[2 * i + 1] = r;
    }
}

/*
 * How many stripe groups, and how many stripes?  A workgroup's time is modelled as stripes x (its waves' tile, + half a
 * sub-window of padding, + RESTAGE entries for the change of stripe) plus its share of the partial sums; the launch lasts
 * as long as its most expensive workgroup.  The two constants are fitted to the kernel as it stands since the packed
 * stream is loaded non-temporally (class C with 10 / 8 / 5 / 4 / 2 groups, four allocations each: 69.8 / 66.8 / 67.1 /
 * 65.8 / 69.1 us, profiles/r03_shapebench_groups_after_nt.log: 1.5 us per stripe staged, 0.84 us per group -- its partial
 * sums are written, read again and added by a second launch); before, a stripe cost twice that and class C ran best with
 * one stripe per workgroup.
 *
 * S stripes rarely divide into the groups: groups of ceil(S/ng) and of floor(S/ng) stripes get row blocks in proportion
 * to their stripes, never fewer than nb_min (the accumulators: a wave carries wave_rows rows at most), and all blocks
 * together must not outnumber the target (a workgroup that does not fit waits for a whole round).  Round 3 found what
 * that costs: on a class D row block of 187 500 rows (92 stripes, 20 groups) the 5-stripe groups could have 13-14 row
 * blocks but the 4-stripe ones were held at 12 by nb_min, so the workgroups of the former streamed 12 % more entries and
 * ended 14 us after the latter (151.5 against 137.9 us of a 161-us launch).  So besides the fewest stripes that fit the
 * LDS buffer, every group count is also tried with the next multiple of itself as the number of (slightly narrower)
 * stripes: 16 groups x 6 stripes x 16 row blocks = 256 equal workgroups for that block.  Of all candidates the one with
 * the cheapest most expensive workgroup wins, the larger group count on a tie.  Class C: 5 groups of 2 stripes (51-52 row
 * blocks each); all of class D: 2 groups of 46 stripes.
 */
#define STRIPE_RESTAGE_ENTRIES 256 /* a change of stripe behind a long tile, in entries of a wave's stream (1.5 us at 174 entries per us) */
#define STRIPE_PARTIAL_FACTOR 2.5  /* a group's partial sums cost 2.5 x their bytes: k_group_sum's own time is in it */
static double geometry_cost(int S, int cand, int nb_min, int target, int wpb, int64_t nnz, int64_t sub, int rows, int elem)
{
    const double per_stripe = (double)nnz / (double)S;
    const int big = (S + cand - 1) / cand, small = S / cand; /* stripes per group */
    const int n_big = S - small * cand, n_small = cand - n_big;
    int nb_big = (int)((int64_t)target * big / S), nb_small = (int)((int64_t)target * small / S);
    nb_big = nb_big < nb_min ? nb_min : nb_big;
    nb_small = nb_small < nb_min ? nb_min : nb_small;
    int left = target - (n_big * nb_big + n_small * nb_small);
    if (left < 0 && cand > 1)
        return -1.0;
    /* what is left of the target: a block more for every group of a kind, the larger groups first */
    if (n_big > 0 && left >= n_big)
        ++nb_big, left -= n_big;
    if (n_small > 0 && left >= n_small)
        ++nb_small, left -= n_small;
    double cost = 0.0;
    for (int kind = 0; kind < 2; ++kind) {
        const int n = kind ? n_small : n_big, st = kind ? small : big, nb = kind ? nb_small : nb_big;
        if (n == 0)
            continue;
        const double tile = per_stripe / ((double)nb * wpb);
        const double padded = tile + 0.5 * (double)sub; /* a tile ends somewhere inside its last sub-window */
        /* a change of stripe hides behind the wave's stream when that is long; behind a tile of a few hundred entries it
         * shows in full: 128 000 rows x 16 (8 stripes) 12.6 / 15.1 / 18.3 us with 1 / 2 / 4 stripes per group, 512 000 x 64
         * (32 stripes) 84 / 90 / 110 us with 4 / 8 / 16 (profiles/r03_shapebench_small_shapes.log) */
        const double restage = STRIPE_RESTAGE_ENTRIES * (1.0 + (tile > 1024.0 ? 1024.0 / tile : 1.0));
        const double c = st * (padded + restage);
        cost = c > cost ? c : cost;
    }
    if (cand > 1) /* the partial sums, in entries (elem + 2 bytes each), per wave; written, read again, and added by a second launch */
        cost += STRIPE_PARTIAL_FACTOR * (double)cand * rows * 2.0 * elem / (elem + 2.0) / ((double)target * wpb);
    return cost;
}

static void choose_geometry(int s_min, int s_cap, int ng_max, int nb_min, int target, int wpb, int64_t nnz, int64_t sub,
                            int rows, int elem, int fixed_stripes, int *ng_out, int *s_out)
{
    int best = ng_max < 1 ? 1 : ng_max, best_s = s_min;
    double best_cost = 1e300;
    int ng_top = target / (nb_min > 0 ? nb_min : 1); /* as many groups as the accumulators allow ... */
    if (ng_top > s_cap)
        ng_top = s_cap;                               /* ... but never more groups than stripes */
    if (ng_top < 1)
        ng_top = 1;
    for (int cand = ng_top; cand >= 1 && cand >= ng_top - 32; --cand) {
        for (int variant = 1; variant >= 0; --variant) { /* on a tie: the stripes that divide into the groups */
            int S = s_min;
            if (variant == 1) {
                if (fixed_stripes)
                    continue;
                S = (s_min + cand - 1) / cand * cand;
                if (S == s_min || S > s_cap)
                    continue;
            }
            if (cand > S)
                continue;
            const double cost = geometry_cost(S, cand, nb_min, target, wpb, nnz, sub, rows, elem);
            if (cost < 0.0)
                continue;
            if (cost < best_cost * (1.0 - 1e-9))
                best_cost = cost, best = cand, best_s = S;
        }
    }
    *ng_out = best, *s_out = best_s;
}

/* stripes, groups, stripe width: everything the layout's shape depends on, from the matrix's dimensions alone (shared by the
 * host builder below and the device builder, spmv_striped_gpu.hip) */
void spmv_striped_plan_geometry(int rows, int cols, int elem_size, int64_t nnz, const spmv_striped_opts *opts, spmv_striped_host *t)
{
    spmv_striped_geometry(rows, cols, elem_size, opts, t);
    if (opts->groups <= 0 && t->S > 1 && rows > 0) {
        const int64_t per_block = (int64_t)opts->wpb * opts->wave_rows;
        int ng = t->NG, ns = t->S;
        /* a stripe narrower than 4/5 of what the buffer holds is not worth the equal division */
        const int s_cap = t->S + t->S / 4 > 32768 ? 32768 : t->S + t->S / 4;
        choose_geometry(t->S, s_cap, t->NG, (int)((rows + per_block - 1) / per_block), opts->target_wgs < 1 ? 1 : opts->target_wgs,
                        opts->wpb, nnz, 64 * (int64_t)t->EPL, rows, elem_size, opts->stripes > 0, &ng, &ns);
        spmv_striped_opts o2 = *opts;
        o2.groups = ng;
        o2.stripes = ns;
        spmv_striped_geometry(rows, cols, elem_size, &o2, t);
    }
}

/*
 * Row blocks per group: in proportion to the group's entries (groups of 5 and of 4 stripes must not get the same number
 * of workgroups: the larger ones would run a quarter longer), at least what the accumulators demand, no more waves than
 * rows; the largest remainders take what is left of the target.  gent[g] = the group's entries, fillers included.
 * Fills gnb[NG], t->n_wg, t->n_ranges, t->NB.
 */
void spmv_striped_alloc_blocks(const spmv_striped_opts *opts, int rows, const int64_t *gent, int *gnb, spmv_striped_host *t)
{
    const int NG = t->NG, WPB = t->WPB;
    int64_t total = 0;
    for (int g = 0; g < NG; ++g)
        total += gent[g];
    const int target = opts->target_wgs < 1 ? 1 : opts->target_wgs;
    const int64_t per_block = (int64_t)WPB * opts->wave_rows;
    const int nb_min = (int)((rows + per_block - 1) / per_block);
    int nb_max = rows / WPB; /* no point in more waves than rows */
    if (nb_max < nb_min)
        nb_max = nb_min;
    int used = 0;
    for (int g = 0; g < NG; ++g) {
        int nb = total > 0 ? (int)((int64_t)target * gent[g] / total) : target / NG;
        nb = nb < nb_min ? nb_min : (nb > nb_max ? nb_max : nb);
        gnb[g] = nb;
        used += nb;
    }
    while (used < target) { /* largest remainder first; ties to the lower group: deterministic */
        int best = -1;
        double best_rem = -1.0;
        for (int g = 0; g < NG; ++g) {
            if (gnb[g] >= nb_max)
                continue;
            const double rem = (total > 0 ? (double)target * (double)gent[g] / (double)total : 0.0) - gnb[g];
            if (rem > best_rem)
                best_rem = rem, best = g;
        }
        if (best < 0)
            break;
        gnb[best]++;
        ++used;
    }
    t->n_wg = used;
    t->n_ranges = used * WPB;
    t->NB = 0;
    for (int g = 0; g < NG; ++g)
        t->NB = gnb[g] > t->NB ? gnb[g] : t->NB;
}

/* the wave ranges (t->rrow) and workgroup descriptors (t->wgd) of all groups from the per-group row weights wgt[NG][rows];
 * both arrays must hold n_ranges * 2 and n_wg * 4 entries */
void spmv_striped_cut_all(const spmv_striped_opts *opts, int rows, const uint32_t *wgt, const int *gnb, spmv_striped_host *t)
{
    const int NG = t->NG, S = t->S, WPB = t->WPB;
    /*
     * XCD bias (opts->xcd_bias, per mille; 0 = none).  The launcher hands workgroups [i * per, (i + 1) * per), per =
     * ceil(n_wg / 8), to XCD i (xcd_contiguous in spmv_kernels.hip), and on every MI355X measured in rounds 2 and 3 the
     * workgroups of XCDs 0, 2, 4, 6 end 2-4 % later than those of XCDs 1, 3, 5, 7 whatever they hold (tools/wgtimes.py).
     * With a bias b the wave ranges of workgroups on even XCDs get (1000 - b) / 1000 of an equal share of their group's
     * weight and those on odd XCDs (1000 + b) / 1000.
     */
    const int per = (t->n_wg + 7) / 8;
    int *share = NULL;
    if (opts->xcd_bias != 0 && t->n_wg >= 16) {
        share = (int *)malloc((size_t)t->NB * WPB * sizeof(int));
    }
    int q = 0;
    for (int g = 0; g < NG; ++g) {
        if (share)
            for (int i = 0; i < gnb[g] * WPB; ++i)
                share[i] = 1000 + ((((q + i / WPB) / per) & 1) ? opts->xcd_bias : -opts->xcd_bias);
        cut_rows(wgt + (size_t)g * rows, rows, gnb[g] * WPB, opts->wave_rows, share, t->rrow + (size_t)q * WPB * 2);
        for (int rb = 0; rb < gnb[g]; ++rb, ++q) {
            t->wgd[4 * q + 0] = (int)((int64_t)g * S / NG);
            t->wgd[4 * q + 1] = (int)((int64_t)(g + 1) * S / NG);
            t->wgd[4 * q + 2] = g;
            t->wgd[4 * q + 3] = rb;
        }
    }
    free(share);
}

/*
 * Tile starts in stream order -- workgroup (g-major), wave, stripe of the group -- from the padded tile sizes
 * tsize[n_ranges][GM]; fills t->tptr, n_packed, n_dummies, max_tile, skewed.  0, or -2: the stream does not fit 32-bit
 * offsets, or more than 1/8 of it would be filler (the matrix is too sparse per stripe for this plan).
 */
int spmv_striped_tile_starts(const int64_t *tsize, const int64_t *rfill, int64_t nnz, spmv_striped_host *t)
{
    const int nr = t->n_ranges, GM = t->GM, WPB = t->WPB;
    const int64_t sub = 64 * (int64_t)t->EPL;
    int64_t total = 0, fillers = 0, longest = 0, longest_wave = 0;
    for (int u = 0; u < nr; ++u) {
        fillers += rfill[u];
        const int32_t *d = t->wgd + (size_t)(u / WPB) * 4;
        const int ns = d[1] - d[0];
        int32_t *tp = t->tptr + (size_t)u * ((size_t)GM + 1);
        const int64_t wave0 = total;
        for (int si = 0; si <= GM; ++si) {
            if (total > 0x7fffffff - 65536)
                return -2;
            tp[si] = (int32_t)total;
            if (si < ns) {
                const int64_t len = tsize[(size_t)u * GM + si];
                total += len;
                longest = len > longest ? len : longest;
            }
        }
        longest_wave = total - wave0 > longest_wave ? total - wave0 : longest_wave;
    }
    t->n_packed = total;
    t->n_dummies = fillers;
    t->max_tile = longest;
    if (fillers * 8 > nnz)
        return -2;
    /*
     * A (row, stripe) segment is never split between waves: one dense row makes its wave's stream many times longer than
     * the others', and the launch lasts as long as that wave needs -- about 1.1 ns per entry (a wave alone streams a
     * 256-entry sub-window in 0.28 us).  That is harmless while the launch would take as long anyway, and still the better
     * plan while it beats the CSR kernels' 8 ps per nonzero on scattered columns (tools/dispatch_audit.py, the heavy-tailed
     * shape: 40 000 rows, one of 20 000 nonzeros among 3.4 M: STRIPED 21.8 us, STREAM 27.7); beyond that -- one wave
     * holding more than 1/137 of all entries, 30 times its share at 4096 waves -- the automatic choice declines.
     * (Round 2 declined from twice the mean TILE, which lost that shape by 27 %.)
     */
    (void)sub;
    t->skewed = (double)longest_wave * 1.1e-9 > (double)nnz * 8.0e-12 && longest_wave > 4 * (total / (nr > 0 ? nr : 1)) + sub;
    return 0;
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
    spmv_striped_plan_geometry(rows, cols, elem_size, (int64_t)rowstr[r1] - rowstr[r0], opts, t);
    const int S = t->S, NG = t->NG, GM = t->GM, WPB = t->WPB;
    if (rows <= 0 || opts->wave_rows < 64 || opts->wave_rows % 64 || S > 32768 /* fastdiv_do */ ||
        (int64_t)NG * rows > ((int64_t)1 << 28))
        return -2;
    const int64_t nnz = (int64_t)rowstr[r1] - rowstr[r0];

    build_ctx c;
    memset(&c, 0, sizeof c);
    c.a = a, c.rowstr = rowstr, c.colidx = colidx;
    c.r0 = r0, c.rows = rows, c.cols = cols, c.elem = elem_size;
    c.S = S, c.W = t->W, c.NG = NG, c.GM = GM, c.WPB = WPB, c.EPL = t->EPL;
    c.G = 16 / elem_size; /* values per 16-byte unit */
    c.cap = t->cap;
    c.SUB = 64 * (int64_t)t->EPL; /* entries per sub-window */
    c.div = fastdiv_make(t->W);
    c.t = t;
    int rc = -1;
    int *gnb = NULL;
    int64_t *gent = NULL;
    c.sg = (int *)malloc((size_t)S * sizeof(int));
    c.gs0 = (int *)malloc(((size_t)NG + 1) * sizeof(int));
    c.wgt = (uint32_t *)malloc((size_t)NG * rows * sizeof(uint32_t));
    gnb = (int *)malloc((size_t)NG * sizeof(int));
    gent = (int64_t *)calloc((size_t)NG, sizeof(int64_t));
    if (!c.sg || !c.gs0 || !c.wgt || !gnb || !gent)
        goto out;
    for (int g = 0; g <= NG; ++g)
        c.gs0[g] = (int)((int64_t)g * S / NG);
    for (int g = 0; g < NG; ++g)
        for (int s = c.gs0[g]; s < c.gs0[g + 1]; ++s)
            c.sg[s] = g;

    spmv_parallel_for(rows, 2048, pass_weigh, &c);
    if (c.bad) {
        rc = c.bad == 1 ? -2 : -1;
        goto out;
    }

    for (int g = 0; g < NG; ++g) {
        const uint32_t *w = c.wgt + (size_t)g * rows;
        int64_t e = 0;
        for (int r = 0; r < rows; ++r)
            e += w[r];
        gent[g] = e;
    }
    spmv_striped_alloc_blocks(opts, rows, gent, gnb, t);
    const int nr = t->n_ranges;
    if ((int64_t)nr * GM > ((int64_t)1 << 26)) {
        rc = -2;
        goto out;
    }
    c.tsize = (int64_t *)malloc((size_t)nr * GM * sizeof(int64_t));
    c.rfill = (int64_t *)calloc((size_t)nr, sizeof(int64_t));
    t->rrow = (int32_t *)calloc((size_t)nr * 2, sizeof(int32_t));
    t->wgd = (int32_t *)calloc((size_t)t->n_wg * 4, sizeof(int32_t));
    t->tptr = (int32_t *)calloc((size_t)nr * ((size_t)GM + 1), sizeof(int32_t));
    if (!c.tsize || !c.rfill || !t->rrow || !t->wgd || !t->tptr)
        goto out;
    spmv_striped_cut_all(opts, rows, c.wgt, gnb, t);
    free(c.wgt);
    c.wgt = NULL;

    spmv_parallel_for(nr, 1, pass_count, &c);
    if (c.bad) {
        rc = c.bad == 1 ? -2 : -1;
        goto out;
    }

    if (spmv_striped_tile_starts(c.tsize, c.rfill, nnz, t) != 0) {
        rc = -2;
        goto out;
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
    free(c.tsize), free(c.rfill), free(c.sg), free(c.gs0), free(c.wgt), free(gnb), free(gent);
    if (rc != 0)
        spmv_striped_free(t);
    return rc;
}
