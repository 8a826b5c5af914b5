/*
 * csrc/spmv_blocked.hip -- the BLOCKED plan: a two-phase y = A*x over (column stripe, row bin) cells, for matrices whose x
 * is far larger than a cache that gathers well and whose rows are too short per stripe for STRIPED (SparseBench's
 * crsmat170u: 4.9 M rows of ~5 uniformly scattered nonzeros, x = 39 MB).
 *
 * The CSR kernels gather x wherever the columns point; when x is ten times an XCD's L2, every gather misses it and
 * fetches a 64-byte line across the fabric for 8 bytes (profiles/traffic.json, sb170/stream: 1.55 GB of fabric reads for
 * 24 M gathers, 4-8 x the algorithmic bytes; 370 us).  STRIPED's answer -- x in LDS, stripe by stripe -- needs every row
 * to have entries in most stripes.  Here NOTHING gathers from memory and nothing is written in pieces smaller than a
 * (stripe, bin) cell:
 *
 *   k_blk_products  phase 1.  The entries are stored in (column stripe, row bin, row) order.  A workgroup keeps ITS stripe
 *                   of x in LDS (up to 160 KB: 256 stripes cover 5.2 M columns of doubles, one per CU), streams packed
 *                   (value, 16-bit stripe-local column) entries, multiplies, and writes the PRODUCTS to where phase 2
 *                   wants them.  Both orders keep a cell contiguous and in the same inner order, so one 4-byte
 *                   destination per FOUR products says where they go (cells are padded to multiples of four).
 *   k_blk_bins      phase 2.  The products in (row bin, column stripe, row) order, next to a 16-bit bin-local row each.  ONE
 *                   wave per bin of 2048 rows streams them and adds every product to its row's sum in the wave's slice of
 *                   LDS (ds_add: the slice belongs to that wave alone), then writes the slice to y.
 *
 * Every array is read as a linear stream (the first round-3 version of this plan gathered x from L2, column block by
 * column block: 422 us on crsmat170u against 371 for STREAM; this one: DESIGN.md 4.7).  The products cross memory once
 * more than in a CSR kernel (8 B written, 8 + 2 B read per entry): the plan pays off only where the gathers cost more
 * than that, which the automatic choice checks (spmv_plan.c).
 *
 * Order of the additions: a row's products are added in the order of the stream (stripe after stripe), 256 entries per
 * step, four instructions of 64 lanes each.  Two lanes of ONE instruction that hold products of the same row are added by
 * the LDS in its own fixed order for a given address pattern -- there is no other wave on these addresses and a wave's
 * LDS operations complete in order --, so the result is a function of the layout (tests compare fresh processes bit for
 * bit), but unlike STRIPED's it is not an order this file can write down.  Not the reference's left-to-right order: parity
 * is the CSR bar of tests/rowtol.py.  One product in flight per matrix (the products are scratch).
 *
 * The layout is built ON THE DEVICE from the CSR copy (k_bb_count, host prefix sums over the cells, k_bb_scatter,
 * k_bb_dest); rows may hold their columns in any order.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "spmv_internal.h"
#include "spmv_wave.h"

namespace {

#define BB_WIN SPMV_BLOCKED_WIN
#define BB_MAX_STRIPES SPMV_BLOCKED_MAX_STRIPES
#define BB_P1_WAVES 16

#define BKCHK(expr)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            spmv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = e_ == hipErrorOutOfMemory ? -3 : -1;                                                \
            goto out;                                                                                \
        }                                                                                            \
    } while (0)

/* elements of x a workgroup of phase 1 keeps in LDS; the last one is the zero column the pads point to */
template <typename T> struct BlkGeom {
    static constexpr int W = sizeof(T) == 8 ? 20480 : 32768; /* 160 KB of doubles; 15-bit columns of floats */
};

/* ---------------------------------------------------------------- phase 1 */
/* workgroup q: stripe wg[3q], windows [wg[3q+1], wg[3q+2]) of the phase-1 stream.  Inside a window the values lie
 * lane-interleaved in 16-byte units (entry i of lane l at (i / E) * 64 E + l E + i % E, as in STRIPED) so that every load
 * is a dense 1-KiB wave access; the columns are linear (8 bytes per lane). */
template <typename T, int U, int ABL = 0>
__global__ void __launch_bounds__(BB_P1_WAVES * 64) k_blk_products(const int *__restrict__ wg, const T *__restrict__ val,
                                                                   const unsigned short *__restrict__ col,
                                                                   const unsigned int *__restrict__ dest4,
                                                                   const T *__restrict__ x, int cols, int wcols,
                                                                   T *__restrict__ prod)
{
    using VT = typename Vec16<T>::val_t;
    constexpr int E = Vec16<T>::E, NV = 4 / E, W = BlkGeom<T>::W, NT = BB_P1_WAVES * 64;
    __shared__ __attribute__((aligned(16))) T s_x[W];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int s = __builtin_amdgcn_readfirstlane(wg[3 * blockIdx.x]);
    const int w0 = __builtin_amdgcn_readfirstlane(wg[3 * blockIdx.x + 1]), w1 = __builtin_amdgcn_readfirstlane(wg[3 * blockIdx.x + 2]);
    const int n = w1 - w0;
    const int a = w0 + (int)((long long)n * w / BB_P1_WAVES), b = w0 + (int)((long long)n * (w + 1) / BB_P1_WAVES);

    VT v[U][NV];
    u32x2 c[U];
    unsigned int d[U];
    auto load = [&](int u, int kk) __attribute__((always_inline)) {
        kk = kk < b ? kk : b - 1; /* beyond the wave's share: its last window again (loaded, not used) */
        kk = kk < w0 ? w0 : kk;   /* a wave without a share (fewer windows than waves) */
        const size_t base = (size_t)kk * BB_WIN;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            v[u][j] = ld<VT, true>(reinterpret_cast<const VT *>(val + base + (size_t)(j * 64 * E)) + lane);
        c[u] = ld<u32x2, true>(reinterpret_cast<const u32x2 *>(col + base) + lane);
        if (ABL & 1) /* timing only (debug build): no destination stream, the products written where they were read */
            d[u] = (unsigned int)(base / 4) + (unsigned int)lane;
        else
            d[u] = __builtin_nontemporal_load(dest4 + base / 4 + lane);
    };
    /* the first U windows are on their way while x is staged */
#pragma unroll
    for (int u = 0; u < U; ++u)
        load(u, a + u);
    {
        const int c0 = s * wcols;
        const int nx = (cols - c0) < wcols ? (cols - c0) : wcols;
        const T *xs = x + c0;
        for (int i = tid * E; i < W; i += NT * E) {
            VT xv;
            if (i + E <= nx) {
                xv = *reinterpret_cast<const VT *>(xs + i);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    xv[e] = i + e < nx ? xs[i + e] : (T)0; /* beyond the stripe: zeros, the zero column W - 1 among them */
            }
            *reinterpret_cast<VT *>(&s_x[i]) = xv;
        }
    }
    __syncthreads();
    for (int k = a; k < b; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            T p[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned int ci = (i & 1) ? (c[u][i >> 1] >> 16) : (c[u][i >> 1] & 0xffffu);
                p[i] = v[u][i / E][i % E] * s_x[ci];
            }
            T *out = prod + (size_t)d[u] * 4;
            load(u, k + u + U);
            if (k + u < b) { /* wave-uniform */
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    VT o;
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        o[e] = p[j * E + e];
                    reinterpret_cast<VT *>(out)[j] = o; /* through the caches: the halves of a line meet in L2 and phase 2 finds the
                                                           products there or in the Infinity Cache (non-temporal: 141 -> 180 us) */
                }
            }
        }
    }
}

/* ---------------------------------------------------------------- phase 2 */
/*
 * N consecutive products of the stream per lane, lanes in stream order: every RUN of products of one row -- a row's entries
 * in a cell are neighbours in the stream -- is summed first (inside the lane, then across lanes with the segmented wave scan
 * of k_striped, started wherever the row changes) and reaches the LDS as ONE addition, by the lane where the run ends.
 * Without this a dense row is a queue: 64 lanes adding to one address take their turns (R-MAT 2^21 by name: 530 us without,
 * 398 with; profiles/r03_blocked_runs.log).  Runs that span two windows are two additions.  Costs 2-4 % where there are no
 * runs (SparseBench 170^3 141-146 -> 146, 130^3 74-76 -> 78), so a layout asks for it only when some row is long enough
 * to have two entries per stripe on average (spmv_dev_blocked.runs).
 */
template <int N>
__device__ __forceinline__ void add_runs(double *__restrict__ sy, const unsigned int (&row)[N], const double (&val)[N], int lane)
{
    const unsigned int head_row = row[0];
    unsigned int cur_row = row[0];
    double cur = val[0], head_val = 0.0;
    int nb = 0; /* row changes inside the lane */
#pragma unroll
    for (int i = 1; i < N; ++i) {
        if (row[i] == cur_row) {
            cur = cur + val[i];
        } else {
            if (nb == 0)
                head_val = cur; /* the lane's first run: may have begun in the lanes below */
            else
                unsafeAtomicAdd(&sy[cur_row], cur); /* begins and ends inside the lane */
            ++nb;
            cur_row = row[i];
            cur = val[i];
        }
    }
    const unsigned int prev_tail = (unsigned int)dpp_mov<0x138, 0xf>((int)cur_row); /* wave_shr:1 */
    const int c = lane != 0 && prev_tail == head_row;                             /* my first run continues the lane below's last */
    double sv = cur;
    int sf = !(nb == 0 && c); /* my last run starts here (a row change inside the lane, or nothing to continue) */
    segscan_step<double, 0x111, 0xf>(sv, sf); /* row_shr:1 */
    segscan_step<double, 0x112, 0xf>(sv, sf); /* row_shr:2 */
    segscan_step<double, 0x114, 0xf>(sv, sf); /* row_shr:4 */
    segscan_step<double, 0x118, 0xf>(sv, sf); /* row_shr:8 */
    segscan_step<double, 0x142, 0xa>(sv, sf); /* row_bcast:15 */
    segscan_step<double, 0x143, 0xc>(sv, sf); /* row_bcast:31 */
    const double below = dpp_mov<0x138, 0xf>(sv);  /* the lane below's last run, summed up to there */
    const int c_next = dpp_mov<0x130, 0xf>(c);     /* wave_shl:1: does the lane above continue my last run?  (lane 63: no) */
    if (nb != 0)
        unsafeAtomicAdd(&sy[head_row], c ? below + head_val : head_val);
    if (!c_next)
        unsafeAtomicAdd(&sy[cur_row], sv);
}

/* One wave per PIECE of the phase-2 stream: windows [pw[i], pw[i+1]), all of bin pb[i] -- rows [b RB, b RB + RB) -- or, for a
 * bin that holds more than its share (dense rows), a run of its stripes; then ps[i] >= 0 names the slot of the piece's
 * partial sums and k_blk_join adds a bin's slots in order.  The stream is linear: lane l takes the products
 * j * 64 E + l E .. + E - 1 of a window with its j-th 16-byte load */
template <typename T, int U, int RB, bool RUNS, int ABL = 0>
__global__ void __launch_bounds__(256) k_blk_bins(const int *__restrict__ pw, const int *__restrict__ pb,
                                                  const int *__restrict__ ps, const T *__restrict__ prod,
                                                  const unsigned short *__restrict__ lrow, int rows, int npieces,
                                                  T *__restrict__ y, double *__restrict__ part)
{
    using VT = typename Vec16<T>::val_t;
    constexpr int E = Vec16<T>::E, NV = 4 / E;
    typedef unsigned short rows_t __attribute__((ext_vector_type(E)));
    /* the sums are doubles for both precisions: ds_add_f32 runs at a fraction of ds_add_f64's rate on gfx950 (SparseBench
     * 160^3 in f32: phase 2 114 us with float slices, 25 with these; profiles/r03_blocked_f32_lds_atomics.log).  A float
     * row is therefore summed in double and rounded once -- closer to the exact sum than any float order, inside the bar */
    __shared__ __attribute__((aligned(16))) double s_y[4][RB];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int piece = blockIdx.x * 4 + w;
    if (piece >= npieces)
        return;
    for (int i = lane; i < RB; i += 64)
        s_y[w][i] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int a = __builtin_amdgcn_readfirstlane(pw[piece]), e = __builtin_amdgcn_readfirstlane(pw[piece + 1]);
    const int b = __builtin_amdgcn_readfirstlane(pb[piece]), slot = __builtin_amdgcn_readfirstlane(ps[piece]);
    if (a < e) {
        VT v[U][NV];
        rows_t r[U][NV];
        auto load = [&](int u, int kk) __attribute__((always_inline)) {
            kk = kk < e ? kk : e - 1;
            const size_t base = (size_t)kk * BB_WIN;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                v[u][j] = reinterpret_cast<const VT *>(prod + base + (size_t)(j * 64 * E))[lane]; /* written a moment ago: may still be cached */
                if (ABL & 2) { /* timing only (debug build): no row stream, every lane a row of its own */
#pragma unroll
                    for (int i = 0; i < E; ++i)
                        r[u][j][i] = (unsigned short)((kk * 61 + j * 64 * E + lane * E + i) % RB);
                } else {
                    r[u][j] = __builtin_nontemporal_load(reinterpret_cast<const rows_t *>(lrow + base + (size_t)(j * 64 * E)) + lane);
                }
            }
        };
#pragma unroll
        for (int u = 0; u < U; ++u)
            load(u, a + u);
        for (int k = a; k < e; k += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                VT q[NV];
                rows_t rr[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    q[j] = v[u][j], rr[j] = r[u][j];
                load(u, k + u + U);
                if (k + u < e) { /* wave-uniform */
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        unsigned int rw[E];
                        double pv[E];
#pragma unroll
                        for (int i = 0; i < E; ++i)
                            rw[i] = rr[j][i], pv[i] = (double)q[j][i];
                        if constexpr (RUNS) {
                            add_runs<E>(s_y[w], rw, pv, lane);
                        } else {
#pragma unroll
                            for (int i = 0; i < E; ++i)
                                unsafeAtomicAdd(&s_y[w][rw[i]], pv[i]);
                        }
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int r0 = b * RB;
    if (slot < 0) {
        for (int i = lane; i < RB && r0 + i < rows; i += 64)
            y[r0 + i] = (T)s_y[w][i];
    } else {
        for (int i = lane; i < RB; i += 64)
            part[(size_t)slot * RB + i] = s_y[w][i];
    }
}

/* workgroup (q, c): rows [256 c, 256 c + 256) of cut bin jn[3 q]; jn[3 q + 1] = its first slot, jn[3 q + 2] = its slots, added in
 * slot order (four loads in flight: a bin of an R-MAT graph's hubs has hundreds of slots) */
template <typename T>
__global__ void __launch_bounds__(256) k_blk_join(const int *__restrict__ jn, const double *__restrict__ part, int rb, int rows,
                                                  T *__restrict__ y)
{
    const int b = jn[3 * blockIdx.x], s0 = jn[3 * blockIdx.x + 1], n = jn[3 * blockIdx.x + 2];
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= rb || b * rb + i >= rows)
        return;
    const double *p = part + (size_t)s0 * rb + i;
    double sum = p[0];
    int k = 1;
    for (; k + 4 <= n; k += 4) {
        const double v0 = p[(size_t)k * rb], v1 = p[(size_t)(k + 1) * rb], v2 = p[(size_t)(k + 2) * rb], v3 = p[(size_t)(k + 3) * rb];
        sum = (((sum + v0) + v1) + v2) + v3;
    }
    for (; k < n; ++k)
        sum = sum + p[(size_t)k * rb];
    y[b * rb + i] = (T)sum;
}

/* ---------------------------------------------------------------- the builder */
/* the row of CSR position j: the last r in [lo, hi] with rowptr[r] <= j */
__device__ __forceinline__ int row_of(const int *__restrict__ rowptr, int lo, int hi, int j)
{
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rowptr[mid] <= j)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

/* entries per (stripe, bin) cell: cnt[s * B + b] */
__global__ void __launch_bounds__(256) k_bb_count(const int *__restrict__ rowptr, const int *__restrict__ col, int rows,
                                                  int nnz, int wcols, int rb, int B, int *__restrict__ cnt)
{
    for (long long jj = (long long)blockIdx.x * 256 + threadIdx.x; jj < nnz; jj += (long long)gridDim.x * 256) {
        const int j = (int)jj;
        const int r = row_of(rowptr, 0, rows - 1, j);
        atomicAdd(&cnt[(size_t)((col[j] - 1) / wcols) * B + r / rb], 1);
    }
}

/* wave per bin: every entry of the bin's rows, in CSR order, to place (rank inside its cell) of both streams.  The rank is
 * counted, not drawn from an atomic: per 64 entries one turn per distinct stripe among them, so the layout is the same
 * in every build. */
template <typename T>
__global__ void __launch_bounds__(256) k_bb_scatter(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                    const T *__restrict__ val, int rows, int rb, int B, int S,
                                                    int wcols, const unsigned int *__restrict__ c1q,
                                                    const unsigned int *__restrict__ c2q, T *__restrict__ val1,
                                                    unsigned short *__restrict__ col1, unsigned short *__restrict__ lrow)
{
    constexpr int E = Vec16<T>::E;
    __shared__ volatile unsigned int s_cur[4][BB_MAX_STRIPES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
        for (int s = lane; s < S; s += 64)
            s_cur[w][s] = 0u;
        __builtin_amdgcn_wave_barrier();
        const int r0 = b * rb;
        const int r1 = (r0 + rb) < rows ? r0 + rb : rows;
        const int e0 = rowptr[r0], e1 = rowptr[r1];
        for (int base = e0; base < e1; base += 64) {
            const int j = base + lane;
            const bool ok = j < e1;
            const int c = ok ? col[j] - 1 : 0;
            const int s = c / wcols;
            const int r = ok ? row_of(rowptr, r0, r1 - 1, j) : r0;
            bool pending = ok;
            unsigned int rank = 0u;
            unsigned long long m;
            while ((m = __ballot(pending)) != 0ull) {
                const int leader = __builtin_ctzll(m);
                const int ks = __shfl(s, leader, 64);
                const bool same = pending && s == ks;
                const unsigned long long mm = __ballot(same);
                const unsigned int first = s_cur[w][ks];
                if (same) {
                    rank = first + (unsigned int)__builtin_popcountll(mm & below);
                    pending = false;
                }
                if (lane == leader)
                    s_cur[w][ks] = first + (unsigned int)__builtin_popcountll(mm);
                __builtin_amdgcn_wave_barrier();
            }
            if (ok) {
                const size_t id = (size_t)s * B + b;
                const size_t p1 = (size_t)c1q[id] * 4 + rank, p2 = (size_t)c2q[id] * 4 + rank;
                const size_t wb = p1 / BB_WIN * BB_WIN;
                const unsigned int o = (unsigned int)(p1 % BB_WIN), l = o / 4, i = o % 4;
                val1[wb + (i / E) * (64 * E) + l * E + (i % E)] = val[j];
                col1[p1] = (unsigned short)(c - s * wcols);
                lrow[p2] = (unsigned short)(r - r0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

/* thread per cell: the destination of each of its groups of four products; what follows the last cell of a stripe up to
 * the window boundary goes to the spare group */
__global__ void __launch_bounds__(256) k_bb_dest(const unsigned int *__restrict__ c1q, const unsigned int *__restrict__ c2q,
                                                 const int *__restrict__ padded, size_t ncells, unsigned int spare,
                                                 unsigned int *__restrict__ dest4)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= ncells)
        return;
    const unsigned int a = c1q[id], n = c1q[id + 1] - a, d = c2q[id], real = (unsigned int)padded[id] / 4u;
    for (unsigned int q = 0; q < n; ++q)
        dest4[a + q] = q < real ? d + q : spare;
}

__global__ void __launch_bounds__(256) k_bb_fill16(unsigned short *__restrict__ p, size_t n, int v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = (unsigned short)v;
}

std::atomic<int> g_blk_bin_rows{0}; /* rows per bin of phase 2 (tuning key 18: 0 = by the row count, or one of the instantiated sizes), read when a layout is built */

} // namespace

extern "C" void spmv_blocked_free(spmv_dev_blocked *d)
{
    (void)hipFree(d->val1), (void)hipFree(d->col1), (void)hipFree(d->dest4), (void)hipFree(d->wg);
    (void)hipFree(d->prod), (void)hipFree(d->lrow), (void)hipFree(d->pw), (void)hipFree(d->pb), (void)hipFree(d->ps);
    (void)hipFree(d->jn), (void)hipFree(d->part);
    memset(d, 0, sizeof *d);
}

extern "C" void spmvk_blocked_set_bin_rows(int bin_rows)
{
    if (bin_rows == 0 || (bin_rows >= 1024 && bin_rows <= 2560 && bin_rows % 256 == 0))
        g_blk_bin_rows.store(bin_rows, std::memory_order_relaxed);
}

/*
 * Builds the BLOCKED layout of a device-resident CSR block (d_rowptr 0-based and local, d_col 1-based).  0; -2: not
 * representable (more than 2048 column stripes, more than 2^24 cells, a stream beyond 2^31 entries) or, for an automatic
 * choice, not suited (see below); -3: out of memory; -1: a HIP error.
 */
extern "C" int spmv_blocked_build_device(const int *d_rowptr, const int *d_col, const void *d_val, int rows, int cols,
                                         int elem_size, int64_t nnz, int auto_choice, void *stream, spmv_dev_blocked *out)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    memset(out, 0, sizeof *out);
    if (rows <= 0 || cols <= 0 || nnz <= 0 || nnz > 0x7ffffff0ll)
        return -2;
    spmv_blocked_plan g;
    if (spmv_blocked_geometry(rows, cols, elem_size, g_blk_bin_rows.load(std::memory_order_relaxed), &g))
        return -2;
    const int S = g.S, wcols = g.wcols, W = g.W, RB = g.RB, B = g.B;
    if (RB < 1024 || RB % 256) /* k_blk_bins is instantiated for 1024, 1280 ... 2560 rows per bin */
        return -2;
    const size_t ncells = (size_t)S * (size_t)B;
    int *d_cnt = nullptr, *h_cnt = nullptr;
    unsigned int *d_c1q = nullptr, *d_c2q = nullptr;
    spmv_dev_blocked d;
    memset(&d, 0, sizeof d);
    h_cnt = (int *)malloc(ncells * sizeof(int));
    if (!h_cnt) {
        rc = -3;
        goto out;
    }
    /* ---- pass 1: entries per cell */
    BKCHK(hipMalloc((void **)&d_cnt, ncells * sizeof(int)));
    BKCHK(hipMemsetAsync(d_cnt, 0, ncells * sizeof(int), st));
    hipLaunchKernelGGL(k_bb_count, dim3(nnz < 256 * 4096 ? (unsigned int)((nnz + 255) / 256) : 4096u), dim3(256), 0, st, d_rowptr,
                       d_col, rows, (int)nnz, wcols, RB, B, d_cnt);
    BKCHK(hipGetLastError());
    BKCHK(hipMemcpyAsync(h_cnt, d_cnt, ncells * sizeof(int), hipMemcpyDeviceToHost, st));
    BKCHK(hipStreamSynchronize(st));
    /* ---- the decisions (spmv_blocked_plan.c): where every cell lies in both streams, workgroups, pieces, cut bins */
    rc = spmv_blocked_plan_build(h_cnt, nnz, auto_choice, &g);
    if (rc == -1)
        spmv_set_error("BLOCKED builder: the entries counted per cell do not add up to the %lld of the matrix", (long long)nnz);
    if (rc)
        goto out;
    {
    const int64_t n1 = g.n1, n2 = g.n2;
    const int n_wg = g.n_wg, n_pieces = g.n_pieces, n_slots = g.n_slots, n_join = g.n_join;
    const unsigned int *h_c1q = g.c1q, *h_c2q = g.c2q;
    const int *h_wg = g.wg, *h_pw = g.pw, *h_pb = g.pb, *h_ps = g.ps, *h_jn = g.jn;
    /* ---- pass 2: the entries to their places */
    BKCHK(hipMalloc((void **)&d_c1q, (ncells + 1) * sizeof(unsigned int)));
    BKCHK(hipMalloc((void **)&d_c2q, ncells * sizeof(unsigned int)));
    BKCHK(hipMalloc(&d.val1, (size_t)n1 * (size_t)elem_size));
    BKCHK(hipMalloc((void **)&d.col1, (size_t)n1 * sizeof(unsigned short)));
    BKCHK(hipMalloc((void **)&d.dest4, (size_t)n1 / 4 * sizeof(unsigned int)));
    BKCHK(hipMalloc(&d.prod, ((size_t)n2 + 4) * (size_t)elem_size)); /* + the spare group */
    BKCHK(hipMalloc((void **)&d.lrow, (size_t)n2 * sizeof(unsigned short)));
    BKCHK(hipMalloc((void **)&d.wg, ((size_t)n_wg + 1) * 3 * sizeof(int)));
    BKCHK(hipMalloc((void **)&d.pw, ((size_t)n_pieces + 1) * sizeof(int)));
    BKCHK(hipMalloc((void **)&d.pb, ((size_t)n_pieces + 1) * sizeof(int)));
    BKCHK(hipMalloc((void **)&d.ps, ((size_t)n_pieces + 1) * sizeof(int)));
    BKCHK(hipMalloc((void **)&d.jn, ((size_t)n_join + 1) * 3 * sizeof(int)));
    BKCHK(hipMalloc((void **)&d.part, ((size_t)n_slots * RB + 1) * sizeof(double)));
    BKCHK(hipMemcpyAsync(d_c1q, h_c1q, (ncells + 1) * sizeof(unsigned int), hipMemcpyHostToDevice, st));
    BKCHK(hipMemcpyAsync(d_c2q, h_c2q, ncells * sizeof(unsigned int), hipMemcpyHostToDevice, st));
    BKCHK(hipMemcpyAsync(d.wg, h_wg, (size_t)n_wg * 3 * sizeof(int), hipMemcpyHostToDevice, st));
    BKCHK(hipMemcpyAsync(d.pw, h_pw, ((size_t)n_pieces + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    BKCHK(hipMemcpyAsync(d.pb, h_pb, (size_t)n_pieces * sizeof(int), hipMemcpyHostToDevice, st));
    BKCHK(hipMemcpyAsync(d.ps, h_ps, (size_t)n_pieces * sizeof(int), hipMemcpyHostToDevice, st));
    if (n_join)
        BKCHK(hipMemcpyAsync(d.jn, h_jn, (size_t)n_join * 3 * sizeof(int), hipMemcpyHostToDevice, st));
    BKCHK(hipMemsetAsync(d.val1, 0, (size_t)n1 * (size_t)elem_size, st));
    BKCHK(hipMemsetAsync(d.prod, 0, ((size_t)n2 + 4) * (size_t)elem_size, st));
    BKCHK(hipMemsetAsync(d.lrow, 0, (size_t)n2 * sizeof(unsigned short), st));
    hipLaunchKernelGGL(k_bb_fill16, dim3(1024), dim3(256), 0, st, d.col1, (size_t)n1, W - 1); /* pads: times the zero column */
    {
        const int grid_b = B < 4 * 4096 ? (B + 3) / 4 : 4096;
        if (elem_size == 8)
            hipLaunchKernelGGL(k_bb_scatter<double>, dim3(grid_b), dim3(256), 0, st, d_rowptr, d_col, (const double *)d_val, rows,
                               RB, B, S, wcols, d_c1q, d_c2q, (double *)d.val1, d.col1, d.lrow);
        else
            hipLaunchKernelGGL(k_bb_scatter<float>, dim3(grid_b), dim3(256), 0, st, d_rowptr, d_col, (const float *)d_val, rows,
                               RB, B, S, wcols, d_c1q, d_c2q, (float *)d.val1, d.col1, d.lrow);
    }
    BKCHK(hipMemcpyAsync(d_cnt, h_cnt, ncells * sizeof(int), hipMemcpyHostToDevice, st)); /* the padded sizes */
    hipLaunchKernelGGL(k_bb_dest, dim3((unsigned int)((ncells + 255) / 256)), dim3(256), 0, st, d_c1q, d_c2q, d_cnt, ncells,
                       (unsigned int)(n2 / 4), d.dest4);
    BKCHK(hipGetLastError());
    BKCHK(hipStreamSynchronize(st));
    d.elem = elem_size, d.rows = rows, d.cols = cols, d.nnz = nnz, d.S = S, d.wcols = wcols, d.W = W, d.RB = RB, d.B = B;
    d.n_wg = n_wg, d.n1 = n1, d.n2 = n2, d.n_pieces = n_pieces, d.n_slots = n_slots, d.n_join = n_join;
    d.bytes = (size_t)n1 * ((size_t)elem_size + 3) + (size_t)n2 * ((size_t)elem_size + 2) + ((size_t)n_wg + 1) * 12 + ((size_t)n_pieces + 1) * 12 +
              ((size_t)n_join + 1) * 12 + ((size_t)n_slots * RB + 1) * 8;
    *out = d;
    memset(&d, 0, sizeof d);
    }
out:
    (void)hipFree(d_cnt), (void)hipFree(d_c1q), (void)hipFree(d_c2q);
    free(h_cnt);
    spmv_blocked_plan_free(&g);
    if (rc != 0) {
        (void)hipGetLastError();
        spmv_blocked_free(&d);
    }
    return rc;
}

/* U = 2 windows in flight per wave in both phases: 1 to 8 differ by less than the boxes do (SparseBench 170^3: 160-164 us with 2,
 * 165-170 with 4, 180 with one window in phase 2; profiles/r03_blocked_prefetch_depth.log): 16 and 8 waves per CU hide the
 * latency, the memory system is what both phases wait for */
#ifdef SPMV_HIP_DEBUG_BUILD
static std::atomic<int> g_blk_abl{0}; /* timing-only ablations (tuning key 26): 1 phase 1 without dest4, 2 phase 2 without lrow */
extern "C" void spmvk_blocked_set_abl(int v) { g_blk_abl.store(v & 3, std::memory_order_relaxed); }
#endif

template <typename T> static void launch_blocked(const spmv_dev_blocked *d, const void *x, void *y, hipStream_t st)
{
    const dim3 g2((d->n_pieces + 3) / 4);
#ifdef SPMV_HIP_DEBUG_BUILD
    const int abl = g_blk_abl.load(std::memory_order_relaxed);
    if (abl && d->RB == 2560 && !d->runs && !d->n_join) { /* the byte-budget probes: SparseBench-like layouts only; results are wrong */
        if (abl & 1)
            hipLaunchKernelGGL((k_blk_products<T, 2, 1>), dim3(d->n_wg), dim3(BB_P1_WAVES * 64), 0, st, d->wg, (const T *)d->val1, d->col1,
                               d->dest4, (const T *)x, d->cols, d->wcols, (T *)d->prod);
        else
            hipLaunchKernelGGL((k_blk_products<T, 2>), dim3(d->n_wg), dim3(BB_P1_WAVES * 64), 0, st, d->wg, (const T *)d->val1, d->col1,
                               d->dest4, (const T *)x, d->cols, d->wcols, (T *)d->prod);
        if (abl & 2)
            hipLaunchKernelGGL((k_blk_bins<T, 2, 2560, false, 2>), g2, dim3(256), 0, st, d->pw, d->pb, d->ps, (const T *)d->prod, d->lrow,
                               d->rows, d->n_pieces, (T *)y, d->part);
        else
            hipLaunchKernelGGL((k_blk_bins<T, 2, 2560, false>), g2, dim3(256), 0, st, d->pw, d->pb, d->ps, (const T *)d->prod, d->lrow,
                               d->rows, d->n_pieces, (T *)y, d->part);
        return;
    }
#endif
    hipLaunchKernelGGL((k_blk_products<T, 2>), dim3(d->n_wg), dim3(BB_P1_WAVES * 64), 0, st, d->wg, (const T *)d->val1, d->col1,
                       d->dest4, (const T *)x, d->cols, d->wcols, (T *)d->prod);
#define BLK_BINS(RB_)                                                                                                          \
    case RB_:                                                                                                                  \
        if (d->runs)                                                                                                           \
            hipLaunchKernelGGL((k_blk_bins<T, 2, RB_, true>), g2, dim3(256), 0, st, d->pw, d->pb, d->ps, (const T *)d->prod, d->lrow, \
                               d->rows, d->n_pieces, (T *)y, d->part);                                                           \
        else                                                                                                                   \
            hipLaunchKernelGGL((k_blk_bins<T, 2, RB_, false>), g2, dim3(256), 0, st, d->pw, d->pb, d->ps, (const T *)d->prod, d->lrow, \
                               d->rows, d->n_pieces, (T *)y, d->part);                                                           \
        break
    switch (d->RB) {
        BLK_BINS(1024);
        BLK_BINS(1280);
        BLK_BINS(1536);
        BLK_BINS(1792);
        BLK_BINS(2048);
        BLK_BINS(2304);
        BLK_BINS(2560);
    }
#undef BLK_BINS
    if (d->n_join)
        hipLaunchKernelGGL(k_blk_join<T>, dim3(d->n_join, (d->RB + 255) / 256), dim3(256), 0, st, d->jn, d->part, d->RB, d->rows,
                           (T *)y);
}

extern "C" int spmvk_blocked(const spmv_dev_blocked *d, const void *x, void *y, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (d->rows <= 0) { /* a plan without its layout (a refresh that failed half-way): never a silent stale y */
        spmv_set_error("BLOCKED: no layout behind the plan");
        return -1;
    }
    if (d->W != (d->elem == 8 ? BlkGeom<double>::W : BlkGeom<float>::W)) { /* the layout was planned for another kernel */
        spmv_set_error("BLOCKED: layout planned for %d elements of x in LDS", d->W);
        return -1;
    }
    if (d->RB < 1024 || d->RB > 2560 || d->RB % 256) { /* the bin sizes phase 2 is instantiated for (launch_blocked's switch) */
        spmv_set_error("BLOCKED: no kernel for bins of %d rows", d->RB);
        return -1;
    }
    if (d->elem == 8)
        launch_blocked<double>(d, x, y, st);
    else
        launch_blocked<float>(d, x, y, st);
    return (int)hipGetLastError();
}
