/*
 * csrc/spmv_plan.c -- host-side planning for the HIP SpMV backend (plain C).
 *
 * Everything here works on the caller's own 1-based `rowstr` array in host
 * memory (the ABI hands it over anyway, /root/reference/libspmv/native.c:3),
 * costs O(rows), and runs once per cached matrix:
 *
 *   - row-length statistics that drive the adaptive kernel choice (the
 *     reference has a precedent for per-matrix dispatch in
 *     libspmv/mkl-model.c:152-165, there between MKL and the native loop);
 *   - the row-block list of the STREAM kernel and the segment list of the
 *     long-row kernels;
 *   - the row bins of the BINNED plan;
 *   - the block-row partition across GPUs.
 *
 * No GPU is needed to run or test any of it.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* sched_getaffinity, CPU_COUNT */
#endif
#include "spmv_internal.h"

#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <unistd.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int row_len(const int *rowstr, int r) { return rowstr[r + 1] - rowstr[r]; }

void spmv_plan_row_stats(const int *rowstr, int r0, int r1, spmv_row_stats *st)
{
    memset(st, 0, sizeof *st);
    st->rows = r1 - r0;
    if (st->rows <= 0) {
        st->rows = 0;
        return;
    }
    st->nnz = (int64_t)rowstr[r1] - (int64_t)rowstr[r0];
    st->min_len = 0x7fffffff;
    double sumsq = 0.0;
    for (int r = r0; r < r1; ++r) {
        int len = row_len(rowstr, r);
        if (len < st->min_len)
            st->min_len = len;
        if (len > st->max_len)
            st->max_len = len;
        if (len <= 0) {
            st->empty_rows++;
            st->hist[0]++;
        } else {
            int k = 1;
            while ((len >> k) != 0)
                ++k; /* 2^(k-1) <= len < 2^k */
            st->hist[k > 32 ? 32 : k]++;
        }
        sumsq += (double)len * (double)len;
    }
    st->mean_len = (double)st->nnz / (double)st->rows;
    double var = sumsq / (double)st->rows - st->mean_len * st->mean_len;
    st->cv_len = (st->mean_len > 0 && var > 0) ? sqrt(var) / st->mean_len : 0.0;
}

/* ---- STREAM row blocks + long-row segments ---------------------------------- */

typedef struct ivec {
    int *p;
    size_t n, cap;
} ivec;

static int ivec_push(ivec *v, int x)
{
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        int *np = (int *)realloc(v->p, nc * sizeof(int));
        if (!np)
            return -1;
        v->p = np;
        v->cap = nc;
    }
    v->p[v->n++] = x;
    return 0;
}

int spmv_plan_stream_build(const int *rowstr, int r0, int r1, int cap_nnz, int cap_rows, int long_seg,
                           spmv_stream_plan *out)
{
    memset(out, 0, sizeof *out);
    ivec blk = {0}, seg = {0}, lrow = {0};
    const int base = rowstr[r0];
    int rows = r1 - r0;
    int bstart = 0;     /* first row of the block being filled (local)        */
    int bnnz = 0;       /* nonzeros in it so far                              */
    int err = 0;

#define PUSH_BLOCK(row)                                                      \
    do {                                                                     \
        err |= ivec_push(&blk, (row));                                       \
        err |= ivec_push(&blk, rowstr[r0 + (row)] - base);                   \
    } while (0)

    if (rows > 0)
        PUSH_BLOCK(0);
    for (int r = 0; r < rows && !err; ++r) {
        int len = row_len(rowstr, r0 + r);
        if (len > cap_nnz) {
            /* close the running block, emit the long row as a block of its own */
            if (r > bstart)
                PUSH_BLOCK(r);
            int nseg = (len + long_seg - 1) / long_seg;
            int s0 = rowstr[r0 + r] - base;
            int slot0 = out->nslots;
            for (int s = 0; s < nseg; ++s) {
                int sb = s0 + s * long_seg;
                int se = sb + long_seg;
                if (se > s0 + len)
                    se = s0 + len;
                err |= ivec_push(&seg, r);
                err |= ivec_push(&seg, sb);
                err |= ivec_push(&seg, se);
                err |= ivec_push(&seg, nseg == 1 ? -1 : out->nslots++);
            }
            if (nseg > 1) {
                err |= ivec_push(&lrow, r);
                err |= ivec_push(&lrow, slot0);
                err |= ivec_push(&lrow, out->nslots);
            }
            out->n_long_rows++;
            bstart = r + 1;
            bnnz = 0;
            if (bstart < rows)
                PUSH_BLOCK(bstart);
            continue;
        }
        if (bnnz + len > cap_nnz || r - bstart >= cap_rows) {
            PUSH_BLOCK(r);
            bstart = r;
            bnnz = 0;
        }
        bnnz += len;
    }
#undef PUSH_BLOCK
    if (err) {
        free(blk.p), free(seg.p), free(lrow.p);
        return -1;
    }
    /* the last pushed pair opened a block; count of blocks = pairs, then sentinel */
    out->nblk = (int)(blk.n / 2);
    if (ivec_push(&blk, rows) || ivec_push(&blk, rows > 0 ? rowstr[r1] - base : 0)) {
        free(blk.p), free(seg.p), free(lrow.p);
        return -1;
    }
    out->blk = blk.p;
    out->nseg = (int)(seg.n / 4);
    out->seg = seg.p;
    out->nlrow = (int)(lrow.n / 3);
    out->lrow = lrow.p;
    return 0;
}

void spmv_plan_stream_free(spmv_stream_plan *p)
{
    free(p->blk), free(p->seg), free(p->lrow);
    memset(p, 0, sizeof *p);
}

/* ---- BINNED plan -------------------------------------------------------------- */

int spmv_plan_bin_of_len(int len)
{
    if (len <= 2)
        return 0;
    if (len <= 16)
        return 1;
    if (len <= 128)
        return 2;
    return 3;
}

int spmv_plan_bin_lanes(int bin)
{
    static const int lanes[SPMV_NBINS] = {1, 4, 16, 64};
    return lanes[bin < 0 ? 0 : (bin >= SPMV_NBINS ? SPMV_NBINS - 1 : bin)];
}

int spmv_plan_bins_build(const int *rowstr, int r0, int r1, int cap_nnz, spmv_bin_plan *out)
{
    memset(out, 0, sizeof *out);
    int rows = r1 - r0;
    int count[SPMV_NBINS] = {0};
    for (int r = 0; r < rows; ++r) {
        int len = row_len(rowstr, r0 + r);
        if (len > cap_nnz)
            continue;
        count[spmv_plan_bin_of_len(len)]++;
    }
    out->bin_off[0] = 0;
    for (int b = 0; b < SPMV_NBINS; ++b)
        out->bin_off[b + 1] = out->bin_off[b] + count[b];
    int total = out->bin_off[SPMV_NBINS];
    out->bin_rows = (int *)malloc((size_t)(total > 0 ? total : 1) * sizeof(int));
    if (!out->bin_rows)
        return -1;
    int cur[SPMV_NBINS];
    for (int b = 0; b < SPMV_NBINS; ++b)
        cur[b] = out->bin_off[b];
    for (int r = 0; r < rows; ++r) { /* stable: row order kept inside a bin */
        int len = row_len(rowstr, r0 + r);
        if (len > cap_nnz)
            continue;
        out->bin_rows[cur[spmv_plan_bin_of_len(len)]++] = r;
    }
    return 0;
}

void spmv_plan_bins_free(spmv_bin_plan *p)
{
    free(p->bin_rows);
    memset(p, 0, sizeof *p);
}

/* ---- host threads for the one-time layout conversions ---------------------------------- */

typedef struct par_task {
    pthread_t th;
    int64_t lo, hi;
    spmv_range_fn fn;
    void *ctx;
} par_task;

static void *par_entry(void *v)
{
    par_task *t = (par_task *)v;
    t->fn(t->lo, t->hi, t->ctx);
    return NULL;
}

/* CPUs this process may use at once: a cgroup CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) when
 * there is one, the affinity mask, the cores online -- whichever is least */
static int cpus_allowed(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0 && a < n)
            n = a;
    }
    long long quota = -1, period = -1;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[32];
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0)
            quota = atoll(q);
        fclose(f);
    } else {
        FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && (fscanf(fq, "%lld", &quota) != 1 || fscanf(fp, "%lld", &period) != 1))
            quota = -1;
        if (fq)
            fclose(fq);
        if (fp)
            fclose(fp);
    }
    if (quota > 0 && period > 0) {
        const long c = (long)((quota + period - 1) / period);
        if (c >= 1 && c < n)
            n = c;
    }
    return n < 1 ? 1 : (int)n;
}

/*
 * Threads of the layout builders.  Their passes scale with the cores of the GPU hosts (NPB class C on an EPYC 9575F:
 * 0.31 / 0.17 / 0.09 / 0.05 s on 4 / 8 / 16 / 32 threads, profiles/r02_upload_breakdown.log); 32 is where the
 * returns end.
 */
int spmv_build_threads(void)
{
    static int n = 0; /* the parts of a row partition build side by side: read and written with atomics; every thread
                         that finds it unset computes the same value */
    int have = __atomic_load_n(&n, __ATOMIC_RELAXED);
    if (have == 0) {
        int want = cpus_allowed();
        if (want > 32)
            want = 32;
        const char *e = getenv("SPMV_HIP_BUILD_THREADS");
        if (e && atoi(e) > 0)
            want = atoi(e);
        have = want < 1 ? 1 : (want > 64 ? 64 : want);
        __atomic_store_n(&n, have, __ATOMIC_RELAXED);
    }
    return have;
}

/* fn(lo, hi, ctx) over [0, n) in contiguous chunks, one per thread; serial when n < 2 * grain or
 * a thread cannot be started.  fn must not depend on how the range was cut. */
void spmv_parallel_for(int64_t n, int64_t grain, spmv_range_fn fn, void *ctx)
{
    int nt = spmv_build_threads();
    if (grain < 1)
        grain = 1;
    if (n / grain < nt)
        nt = (int)(n / grain);
    if (nt <= 1) {
        if (n > 0)
            fn(0, n, ctx);
        return;
    }
    par_task task[64];
    int started = 0;
    for (int i = 0; i < nt; ++i) {
        task[i].lo = n * i / nt;
        task[i].hi = n * (i + 1) / nt;
        task[i].fn = fn;
        task[i].ctx = ctx;
    }
    for (int i = 1; i < nt; ++i) {
        if (pthread_create(&task[i].th, NULL, par_entry, &task[i]) != 0)
            break;
        started = i;
    }
    fn(task[0].lo, task[0].hi, ctx);
    for (int i = 1; i <= started; ++i)
        pthread_join(task[i].th, NULL);
    for (int i = started + 1; i < nt; ++i) /* threads that could not be started: do their share here */
        fn(task[i].lo, task[i].hi, ctx);
}

/* ---- adaptive choice ------------------------------------------------------------ */

void spmv_plan_choose(const spmv_row_stats *st, int elem_size, int cols, int *plan, int *param)
{
    /*
     * Default: the STREAM row-segment kernel, which reads a/colidx with the same coalesced
     * pattern whatever the row structure (5.8 TB/s algorithmic on MI355X when the gathers of x
     * are local, profiles/r01_kbench2_*.log).
     */
    *plan = SPMV_HIP_PLAN_STREAM;
    *param = SPMV_STREAM_CAP;
    if (st->rows <= 0)
        return;
    /*
     * Tiny matrices (a few dozen row blocks at most) are pure launch latency; one wavefront per
     * row has no LDS phase and no barrier and finishes in 2.6 us where STREAM takes 4.0
     * (NPB class S; profiles/r01_kbench20_dispatch_table.log).  Not with a row long enough to
     * make one wave loop for microseconds.
     */
    if (st->nnz <= 131072 && st->rows <= 8192 && st->max_len <= 4096) {
        *plan = SPMV_HIP_PLAN_VECTOR;
        *param = st->mean_len >= 32.0 || st->max_len > 512 ? 64 : (st->mean_len >= 12.0 ? 16 : 4);
        if (st->rows <= 2048 && st->mean_len >= 12.0)
            *param = 64; /* so few rows that lanes, not rows, fill the chip: 2000 x 16: 2.7 against 3.0 us (r03_dispatch_audit) */
        return;
    }
    /*
     * STRIPED (2-D: one stripe of x resident in LDS per workgroup) when the gathers would
     * otherwise miss: x much larger than a stripe is not required, but every (row, stripe) segment
     * must hold a nonzero or two on average, or the row-end markers and per-stripe partial sums cost
     * more than the gathers they save (512 000 rows x 64 over 32 stripes, 2 per segment: 127 us
     * against 236 us for STREAM; below ~1.7 per segment the builder's filler limit declines).  Matrices whose columns are local to the rows (banded)
     * fail the builder's filler test and stay on STREAM.  Measured on NPB class C: STREAM 224 us,
     * TILED 122 us (bit-exact), STRIPED 80 us; worth it from about 2 M nonzeros.
     */
    const int W = SPMV_STRIPE_XBYTES / elem_size - 16 / elem_size; /* spmv_striped_geometry */
    const int S = cols > 0 ? (cols + W - 1) / W : 1;
    if (st->nnz >= 2000000 && st->rows >= 1024 && cols >= W / 2 && S <= 128 && st->mean_len / (double)S >= 1.6) {
        *plan = SPMV_HIP_PLAN_STRIPED; /* 20 000 rows x 250 nonzeros: 13 us against 26 us for STREAM */
        *param = W;
        return;
    }
    /*
     * BLOCKED (two phases over (column stripe, row bin) cells: x in LDS, products binned by row, csrc/spmv_blocked.hip) when
     * x is beyond what gathers well -- an XCD's L2 holds 4 MB -- and the rows are too short for STRIPED.  SparseBench n^3
     * against STREAM (profiles/r03_dispatch_audit.log): 170^3 (x = 39 MB, 4.9 entries per row) 371 us -> BLOCKED, 160^3 123
     * against 300, 140^3 88 / 187, 120^3 65 / 107, 100^3 (8 MB) 44 / 52; 2 M rows x 16 scattered entries 169 / 482; but
     * 512 000 rows x 4 (x = 4 MB) 41 / 30 the other way.  Whether the columns really are scattered -- a banded matrix of this
     * shape gathers from cache: 2.5 M rows x 5, 74 / 30 -- and the rows even -- an R-MAT graph piles its hubs into one bin:
     * 2^19 vertices 914 / 56 -- is for the builder to see (it counts the entries of every cell) and to decline.  In floats the
     * plan is cheaper still (17 instead of 30 bytes per entry through memory) but so are the gathers (16 elements per line):
     * 120^3 44 / 70, 100^3 32 / 30: from 1.25 M columns (5 MB) on, against 0.94 M columns (7.5 MB) of doubles.
     */
    if (cols >= (elem_size == 8 ? 983040 : 1310720) && st->nnz >= 4000000 && st->rows >= 262144 && st->max_len <= 1048576) {
        *plan = SPMV_HIP_PLAN_BLOCKED;
        *param = 0;
        return;
    }
    /*
     * Still small (a few microseconds of kernel) and the rows are alike: lane groups straight from
     * global memory beat the two-phase STREAM kernel by 10-15 % (NPB class W 3.7 vs 4.3 us, class A
     * 8.3 vs 9.9 us).  Not for skewed rows: on R-MAT 2^14 a 16-lane group per row takes 27 us where
     * STREAM takes 10.
     */
    if (st->nnz <= 4000000 && st->rows <= (st->mean_len < 12.0 ? 131072 : 32768) &&
        st->max_len <= 4.0 * st->mean_len + 16.0) {
        *plan = SPMV_HIP_PLAN_VECTOR;
        *param = st->mean_len < 12.0 ? 4 : (st->rows <= 8192 && st->mean_len >= 32.0 ? 64 : 16);
        return;
    }
}

/* ---- multi-GPU block-row partition ------------------------------------------------ */

void spmv_hip_partition_rows(const int *rowstr, int rows, int parts, int *bounds)
{
    if (parts < 1)
        parts = 1;
    bounds[0] = 0;
    if (rows <= 0) {
        for (int p = 1; p <= parts; ++p)
            bounds[p] = 0;
        return;
    }
    const int64_t base = rowstr[0];
    const int64_t nnz = (int64_t)rowstr[rows] - base;
    for (int p = 1; p < parts; ++p) {
        /* first row whose starting offset reaches p/parts of the nonzeros */
        int64_t target = (nnz * p) / parts;
        int lo = bounds[p - 1], hi = rows;
        while (lo < hi) {
            int mid = lo + (hi - lo) / 2;
            if ((int64_t)rowstr[mid] - base < target)
                lo = mid + 1;
            else
                hi = mid;
        }
        bounds[p] = lo;
    }
    bounds[parts] = rows;
    if (nnz == 0) /* no weights to balance: split rows evenly */
        for (int p = 1; p < parts; ++p)
            bounds[p] = (int)(((int64_t)rows * p) / parts);
}
