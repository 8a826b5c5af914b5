/*
 * csrc/spmv_blocked.hip -- the BLOCKED plan: column-blocked y = A*x for matrices whose x does not fit a cache that
 * gathers well and whose rows are too short per stripe for STRIPED (SparseBench's crsmat170u: 4.9 M rows of ~5 uniformly
 * scattered nonzeros, x = 39 MB).
 *
 * The CSR kernels gather x wherever the columns point; when x is ten times an XCD's L2, every gather misses it and
 * fetches a 64-byte line across the fabric for 8 bytes (profiles/traffic.json, sb170/stream: 1.55 GB of fabric reads for
 * 24 M gathers, 4-8 x the algorithmic bytes; 431 us).  Here the entries are re-ordered BLOCK-MAJOR -- column block c =
 * columns [c * CB, (c + 1) * CB), 2 MB of x, inside a block row after row, inside a row in CSR order -- and XCD c % 8 works
 * on block c: everything an XCD gathers at one time comes out of one x block that fits its 4-MB L2.
 *
 *   k_blk_sums   phase 1: waves stream units of ~2048 entries of their XCD's blocks (value, column with the "last entry
 *                of its (row, block) segment" flag in bit 31), multiply by x, and sum every segment with the segmented
 *                wave scan of k_striped (DPP; a carry joins the groups of 64 a segment spans); segment sums go to a
 *                compact array, one per segment, in stream order
 *   k_blk_rows   phase 2: a wave per 64 rows; lane l adds the segment sums of row 64 g + l, block after block: which
 *                rows of the group have a segment in a block is one 64-bit mask per (block, row group), so a lane finds
 *                its sum by counting bits -- no per-row index lists
 *
 * Deterministic (fixed trees, fixed block order, no atomics on the data path), not the reference's left-to-right order:
 * parity is the CSR bar of tests/rowtol.py.  One product in flight per matrix (the segment sums are scratch).
 *
 * The layout is built ON THE DEVICE from the CSR copy (k_bk_cells, a scan, k_bk_scatter).  With up to 64 column blocks
 * (x up to 128 MB) the rows' columns may come in any order; beyond that rows must be sorted by column (then a
 * (row, block) segment is one run of the row).
 */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "spmv_internal.h"
#include "spmv_wave.h"

namespace {

#define BK_MAX_BLOCKS 512
#define BK_UNIT 2048 /* entries per work unit of phase 1 */

#define BKCHK(expr)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            spmv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = e_ == hipErrorOutOfMemory ? -3 : -1;                                                \
            goto out;                                                                                \
        }                                                                                            \
    } while (0)

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

/* ---------------------------------------------------------------- phase 1 */
template <typename T, int U>
__global__ void __launch_bounds__(256) k_blk_sums(const T *__restrict__ bval, const unsigned int *__restrict__ bcol,
                                                  const int *__restrict__ unit_cell, const int *__restrict__ blk_unit0,
                                                  const int *__restrict__ cell_ent, const int *__restrict__ cell_seg,
                                                  int nblk, const T *__restrict__ x, T *__restrict__ seg_sum)
{
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;
    const int wave_in_xcd = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int waves_per_xcd = (gridDim.x >> 3) * 4;
    for (int b = xcd; b < nblk; b += 8) {
        const int u1 = blk_unit0[b + 1];
        for (int u = blk_unit0[b] + wave_in_xcd; u < u1; u += waves_per_xcd) {
            const int c0 = unit_cell[2 * u], c1 = unit_cell[2 * u + 1]; /* first and end cell of the unit */
            const int e0 = cell_ent[c0], e1 = cell_ent[c1];
            int s = cell_seg[c0];
            T carry = (T)0; /* open sum of the segment that spans the boundary of two groups of 64 */
            for (int eb = e0; eb < e1; eb += 64 * U) {
                T a[U];
                unsigned int ci[U];
#pragma unroll
                for (int k = 0; k < U; ++k) { /* every load of U groups, then every gather, before the first scan */
                    const int me = eb + k * 64 + lane;
                    const bool ok = me < e1;
                    a[k] = ok ? bval[me] : (T)0;
                    ci[k] = ok ? bcol[me] : 0u;
                }
                T v[U];
#pragma unroll
                for (int k = 0; k < U; ++k)
                    v[k] = a[k] * x[ci[k] & 0x7fffffffu];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (eb + k * 64 >= e1) /* wave-uniform */
                        break;
                    const bool ok = eb + k * 64 + lane < e1;
                    const int fend = ok ? (int)(ci[k] >> 31) : 0;
                    /* a lane starts a segment when the lane below it ended one; lane 0 when no sum is open */
                    int fstart = dpp_mov<0x138, 0xf>(fend); /* wave_shr:1 */
                    if (lane == 0)
                        fstart = 0; /* the open sum (carry) is added below where no start precedes the lane */
                    T sv = ok ? v[k] : (T)0;
                    int sf = fstart;
                    segscan_step<T, 0x111, 0xf>(sv, sf); /* row_shr:1 */
                    segscan_step<T, 0x112, 0xf>(sv, sf); /* row_shr:2 */
                    segscan_step<T, 0x114, 0xf>(sv, sf); /* row_shr:4 */
                    segscan_step<T, 0x118, 0xf>(sv, sf); /* row_shr:8 */
                    segscan_step<T, 0x142, 0xa>(sv, sf); /* row_bcast:15 */
                    segscan_step<T, 0x143, 0xc>(sv, sf); /* row_bcast:31 */
                    if (!sf)
                        sv = sv + carry; /* no segment start at or before this lane: the previous group's open sum */
                    const unsigned long long ends = __ballot(fend != 0);
                    if (fend) {
                        const unsigned int before = __builtin_amdgcn_mbcnt_hi((unsigned int)(ends >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((unsigned int)ends, 0));
                        seg_sum[s + (int)before] = sv;
                    }
                    s += __builtin_popcountll(ends);
                    const T last = __shfl(sv, 63, 64);
                    carry = (ends >> 63) ? (T)0 : last; /* lanes beyond e1 hold the last valid lane's sum (they add 0) */
                }
            }
        }
    }
}

/* ---------------------------------------------------------------- phase 2 */
template <typename T>
__global__ void __launch_bounds__(256) k_blk_rows(const unsigned long long *__restrict__ cell_mask,
                                                  const int *__restrict__ cell_seg, const T *__restrict__ seg_sum,
                                                  T *__restrict__ y, int rows, int nrg, int nblk)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long below = lanes_below(lane);
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g < nrg; g += gridDim.x * 4) {
        T acc = (T)0;
        int b = 0;
        for (; b + 4 <= nblk; b += 4) { /* four blocks' loads in flight, added in block order */
            unsigned long long m[4];
            int base[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t i = (size_t)(b + k) * nrg + g;
                m[k] = cell_mask[i];
                base[k] = cell_seg[i];
            }
            T v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = (m[k] >> lane) & 1ull ? seg_sum[base[k] + __builtin_popcountll(m[k] & below)] : (T)0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((m[k] >> lane) & 1ull)
                    acc = acc + v[k];
        }
        for (; b < nblk; ++b) {
            const size_t i = (size_t)b * nrg + g;
            const unsigned long long m = cell_mask[i];
            if ((m >> lane) & 1ull)
                acc = acc + seg_sum[cell_seg[i] + __builtin_popcountll(m & below)];
        }
        const int r = g * 64 + lane;
        if (r < rows)
            y[r] = acc;
    }
}

/* ---------------------------------------------------------------- the builder */
/* wave per row group: entries, segments and row mask of every (block, row group) cell; lane l walks row 64 g + l */
__global__ void __launch_bounds__(256) k_bk_cells(const int *__restrict__ rowptr, const int *__restrict__ col, int rows,
                                                  int nrg, int nblk, int CB, int *__restrict__ cell_ent,
                                                  int *__restrict__ cell_seg, unsigned long long *__restrict__ cell_mask,
                                                  int *__restrict__ flags)
{
    __shared__ int s_e[4][BK_MAX_BLOCKS];
    __shared__ int s_s[4][BK_MAX_BLOCKS];
    __shared__ unsigned long long s_m[4][BK_MAX_BLOCKS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bad_order = 0;
    for (int g = blockIdx.x * 4 + w; g < nrg; g += gridDim.x * 4) {
        for (int b = lane; b < nblk; b += 64)
            s_e[w][b] = 0, s_s[w][b] = 0, s_m[w][b] = 0ull;
        __builtin_amdgcn_wave_barrier();
        const int r = g * 64 + lane;
        if (r < rows && nblk <= 64) {
            /* up to 64 blocks: the blocks a row touches fit one 64-bit mask, so its columns may come in any order (the
             * SparseBench matrix as the Fortran reader sees it, SURVEY quirk N1, has rows that are not sorted) */
            unsigned long long touched = 0ull;
            int prev = -1;
            for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) {
                const int c = col[p] - 1;
                bad_order |= c < prev;
                prev = c;
                atomicAdd(&s_e[w][c / CB], 1);
                touched |= 1ull << (c / CB);
            }
            while (touched) {
                const int b = __builtin_ctzll(touched);
                touched &= touched - 1ull;
                atomicAdd(&s_s[w][b], 1);
                atomicOr(&s_m[w][b], 1ull << lane);
            }
        } else if (r < rows) {
            int p = rowptr[r];
            const int e = rowptr[r + 1];
            int prev = -1;
            while (p < e) {
                const int c = col[p] - 1;
                const int b = c / CB;
                bad_order |= c < prev;
                int n = 1;
                prev = c;
                while (p + n < e) {
                    const int c2 = col[p + n] - 1;
                    if (c2 / CB != b)
                        break;
                    bad_order |= c2 < prev;
                    prev = c2;
                    ++n;
                }
                atomicAdd(&s_e[w][b], n);
                atomicAdd(&s_s[w][b], 1);
                atomicOr(&s_m[w][b], 1ull << lane);
                p += n;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int b = lane; b < nblk; b += 64) {
            const size_t i = (size_t)b * nrg + g;
            cell_ent[i] = s_e[w][b];
            cell_seg[i] = s_s[w][b];
            cell_mask[i] = s_m[w][b];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (bad_order)
        flags[0] = 1;
}

/* exclusive scan of two int arrays in pieces of 2048: in place, the piece totals to tot[2 * piece + {0, 1}] */
__global__ void __launch_bounds__(256) k_bk_scan_local(int *__restrict__ a0, int *__restrict__ a1, size_t n,
                                                       long long *__restrict__ tot)
{
    __shared__ int s_w[2][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t base = (size_t)blockIdx.x * 2048 + (size_t)tid * 8;
    int v0[8], v1[8], sum0 = 0, sum1 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v0[i] = base + i < n ? a0[base + i] : 0;
        v1[i] = base + i < n ? a1[base + i] : 0;
        sum0 += v0[i], sum1 += v1[i];
    }
    int inc0 = sum0, inc1 = sum1;
    for (int o = 1; o < 64; o <<= 1) {
        const int t0 = __shfl_up(inc0, o, 64), t1 = __shfl_up(inc1, o, 64);
        if (lane >= o)
            inc0 += t0, inc1 += t1;
    }
    if (lane == 63)
        s_w[0][w] = inc0, s_w[1][w] = inc1;
    __syncthreads();
    int b0 = inc0 - sum0, b1 = inc1 - sum1;
    for (int k = 0; k < w; ++k)
        b0 += s_w[0][k], b1 += s_w[1][k];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (base + i < n)
            a0[base + i] = b0, a1[base + i] = b1;
        b0 += v0[i], b1 += v1[i];
    }
    if (tid == 0) {
        tot[2 * (size_t)blockIdx.x] = (long long)s_w[0][0] + s_w[0][1] + s_w[0][2] + s_w[0][3];
        tot[2 * (size_t)blockIdx.x + 1] = (long long)s_w[1][0] + s_w[1][1] + s_w[1][2] + s_w[1][3];
    }
}

/* adds the pieces' offsets (off[2 * piece + {0,1}]); entry n of each array = the grand total */
__global__ void __launch_bounds__(256) k_bk_scan_add(int *__restrict__ a0, int *__restrict__ a1, size_t n,
                                                     const long long *__restrict__ off, long long total0, long long total1)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        a0[i] += (int)off[2 * (i / 2048)];
        a1[i] += (int)off[2 * (i / 2048) + 1];
    } else if (i == n) {
        a0[n] = (int)total0;
        a1[n] = (int)total1;
    }
}

/* wave per row group: block after block, every lane's run of the block goes behind the runs of the lanes below it */
template <typename T, bool SORTED>
__global__ void __launch_bounds__(256) k_bk_scatter(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                    const T *__restrict__ val, int rows, int nrg, int nblk, int CB,
                                                    const int *__restrict__ cell_ent,
                                                    const unsigned long long *__restrict__ cell_mask,
                                                    T *__restrict__ bval, unsigned int *__restrict__ bcol)
{
    const int lane = threadIdx.x & 63;
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g < nrg; g += gridDim.x * 4) {
        const int r = g * 64 + lane;
        int p = r < rows ? rowptr[r] : 0;
        const int e = r < rows ? rowptr[r + 1] : 0;
        for (int b = 0; b < nblk; ++b) {
            const size_t i = (size_t)b * nrg + g;
            if (cell_mask[i] == 0ull) /* wave-uniform: no row of the group has an entry in this block */
                continue;
            int n = 0;
            if constexpr (SORTED) {
                while (p + n < e && (col[p + n] - 1) / CB == b)
                    ++n;
            } else { /* the row's entries of this block lie anywhere in the row: count them, keep their order */
                for (int q = p; q < e; ++q)
                    n += (col[q] - 1) / CB == b;
            }
            int inc = n;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, 64);
                if (lane >= o)
                    inc += t;
            }
            const int dst = cell_ent[i] + inc - n;
            if constexpr (SORTED) {
                for (int k = 0; k < n; ++k) {
                    bval[dst + k] = val[p + k];
                    bcol[dst + k] = (unsigned int)(col[p + k] - 1) | (k == n - 1 ? 0x80000000u : 0u);
                }
                p += n;
            } else {
                int k = 0;
                for (int q = p; q < e && k < n; ++q)
                    if ((col[q] - 1) / CB == b) {
                        bval[dst + k] = val[q];
                        bcol[dst + k] = (unsigned int)(col[q] - 1) | (k == n - 1 ? 0x80000000u : 0u);
                        ++k;
                    }
            }
        }
    }
}

} // namespace

extern "C" void spmv_blocked_free(spmv_dev_blocked *d)
{
    (void)hipFree(d->bval), (void)hipFree(d->bcol), (void)hipFree(d->cell_ent), (void)hipFree(d->cell_seg);
    (void)hipFree(d->cell_mask), (void)hipFree(d->unit_cell), (void)hipFree(d->blk_unit0), (void)hipFree(d->seg_sum);
    memset(d, 0, sizeof *d);
}

/*
 * Builds the BLOCKED layout of a device-resident CSR block (d_rowptr 0-based and local, d_col 1-based).  0; -2: not
 * representable (more than 512 column blocks; rows not sorted by column when there are more than 64); -3: out of memory; -1: a HIP error.
 */
extern "C" int spmv_blocked_build_device(const int *d_rowptr, const int *d_col, const void *d_val, int rows, int cols,
                                         int elem_size, int64_t nnz, void *stream, spmv_dev_blocked *out)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    memset(out, 0, sizeof *out);
    const int CB = SPMV_BLOCKED_XBYTES / elem_size;
    const int nblk = cols > 0 ? (cols + CB - 1) / CB : 1;
    const int nrg = (rows + 63) / 64;
    if (rows <= 0 || nnz <= 0 || nblk > BK_MAX_BLOCKS || nnz > 0x7ffffff0ll)
        return -2;
    const size_t nc = (size_t)nblk * nrg;
    const size_t npieces = (nc + 2047) / 2048;
    int *d_flags = nullptr;
    long long *d_tot = nullptr, *h_tot = nullptr;
    int *h_ent = nullptr, *unit_cell = nullptr, *blk_unit0 = nullptr;
    int flag = 0;
    spmv_dev_blocked d;
    memset(&d, 0, sizeof d);
    const int grid_g = nrg < 4 * 8192 ? (nrg + 3) / 4 : 8192;

    BKCHK(hipMalloc((void **)&d.cell_ent, (nc + 1) * sizeof(int)));
    BKCHK(hipMalloc((void **)&d.cell_seg, (nc + 1) * sizeof(int)));
    BKCHK(hipMalloc((void **)&d.cell_mask, nc * sizeof(unsigned long long)));
    BKCHK(hipMalloc((void **)&d_flags, sizeof(int)));
    BKCHK(hipMalloc((void **)&d_tot, 2 * npieces * sizeof(long long)));
    h_tot = (long long *)malloc(2 * npieces * sizeof(long long));
    h_ent = (int *)malloc((nc + 1) * sizeof(int));
    blk_unit0 = (int *)malloc(((size_t)nblk + 1) * sizeof(int));
    if (!h_tot || !h_ent || !blk_unit0) {
        rc = -3;
        goto out;
    }
    BKCHK(hipMemsetAsync(d_flags, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_bk_cells, dim3(grid_g), dim3(256), 0, st, d_rowptr, d_col, rows, nrg, nblk, CB, d.cell_ent, d.cell_seg,
                       d.cell_mask, d_flags);
    hipLaunchKernelGGL(k_bk_scan_local, dim3((unsigned int)npieces), dim3(256), 0, st, d.cell_ent, d.cell_seg, nc, d_tot);
    BKCHK(hipGetLastError());
    BKCHK(hipMemcpyAsync(&flag, d_flags, sizeof(int), hipMemcpyDeviceToHost, st));
    BKCHK(hipMemcpyAsync(h_tot, d_tot, 2 * npieces * sizeof(long long), hipMemcpyDeviceToHost, st));
    BKCHK(hipStreamSynchronize(st));
    if (flag && nblk > 64) { /* a row's columns decrease: its (row, block) segments are not runs, and with more than 64
                                blocks the general path (one block mask per row) does not apply */
        rc = -2;
        goto out;
    }
    {
        long long run0 = 0, run1 = 0;
        for (size_t k = 0; k < npieces; ++k) {
            const long long t0 = h_tot[2 * k], t1 = h_tot[2 * k + 1];
            h_tot[2 * k] = run0, h_tot[2 * k + 1] = run1;
            run0 += t0, run1 += t1;
        }
        if (run0 != nnz || run1 > 0x7ffffff0ll) {
            spmv_set_error("BLOCKED builder: %lld entries counted, %lld expected", run0, (long long)nnz);
            rc = -1;
            goto out;
        }
        d.nseg = (int64_t)run1;
        BKCHK(hipMemcpyAsync(d_tot, h_tot, 2 * npieces * sizeof(long long), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_bk_scan_add, dim3((unsigned int)((nc + 1 + 255) / 256)), dim3(256), 0, st, d.cell_ent, d.cell_seg, nc,
                           d_tot, run0, run1);
        BKCHK(hipGetLastError());
    }
    /* the entries to their places, while the host cuts the work units from the cells' entry counts */
    {
        const size_t n_alloc = (size_t)nnz + 64 * 8;
        BKCHK(hipMalloc(&d.bval, n_alloc * (size_t)elem_size));
        BKCHK(hipMalloc((void **)&d.bcol, n_alloc * sizeof(unsigned int)));
        BKCHK(hipMalloc(&d.seg_sum, ((size_t)d.nseg + 64) * (size_t)elem_size));
        BKCHK(hipMemsetAsync((char *)d.bval + (size_t)nnz * elem_size, 0, 64 * 8 * (size_t)elem_size, st));
        BKCHK(hipMemsetAsync(d.bcol + nnz, 0, 64 * 8 * sizeof(unsigned int), st));
#define BK_SCATTER(T, SORTED)                                                                                                  \
    hipLaunchKernelGGL((k_bk_scatter<T, SORTED>), dim3(grid_g), dim3(256), 0, st, d_rowptr, d_col, (const T *)d_val, rows, nrg, \
                       nblk, CB, d.cell_ent, d.cell_mask, (T *)d.bval, d.bcol)
        if (elem_size == 8) {
            if (flag)
                BK_SCATTER(double, false);
            else
                BK_SCATTER(double, true);
        } else {
            if (flag)
                BK_SCATTER(float, false);
            else
                BK_SCATTER(float, true);
        }
#undef BK_SCATTER
        BKCHK(hipGetLastError());
        BKCHK(hipMemcpyAsync(h_ent, d.cell_ent, (nc + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
        BKCHK(hipStreamSynchronize(st));
    }
    {
        /* units: runs of whole cells of ONE block holding about BK_UNIT entries (a cell is never split: a segment
         * belongs to one unit, so no sum crosses units; a cell with a long row makes a long unit); stored as
         * (first cell, end cell) pairs */
        const size_t cap = (size_t)(nnz / BK_UNIT) + (size_t)nblk + 2;
        size_t nu = 0;
        unit_cell = (int *)malloc(cap * 2 * sizeof(int));
        if (!unit_cell) {
            rc = -3;
            goto out;
        }
        for (int b = 0; b < nblk; ++b) {
            blk_unit0[b] = (int)nu;
            size_t c = (size_t)b * nrg;
            const size_t cend = (size_t)(b + 1) * nrg;
            while (c < cend && h_ent[c] < h_ent[cend] && nu < cap) { /* entries left in the block */
                const long long target = (long long)h_ent[c] + BK_UNIT;
                unit_cell[2 * nu] = (int)c;
                ++c;
                while (c < cend && h_ent[c] < target)
                    ++c;
                unit_cell[2 * nu + 1] = (int)c;
                ++nu;
            }
        }
        blk_unit0[nblk] = (int)nu;
        d.n_units = (int)nu;
        BKCHK(hipMalloc((void **)&d.unit_cell, (nu + 1) * 2 * sizeof(int)));
        BKCHK(hipMalloc((void **)&d.blk_unit0, ((size_t)nblk + 1) * sizeof(int)));
        BKCHK(hipMemcpyAsync(d.unit_cell, unit_cell, nu * 2 * sizeof(int), hipMemcpyHostToDevice, st));
        BKCHK(hipMemcpyAsync(d.blk_unit0, blk_unit0, ((size_t)nblk + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        BKCHK(hipStreamSynchronize(st));
    }
    d.elem = elem_size, d.rows = rows, d.cols = cols, d.nnz = nnz, d.nblk = nblk, d.nrg = nrg, d.CB = CB;
    d.bytes = (nc + 1) * 8 + nc * 8 + ((size_t)nnz + 512) * ((size_t)elem_size + 4) + ((size_t)d.nseg + 64) * (size_t)elem_size +
              ((size_t)d.n_units + 1) * 8 + ((size_t)nblk + 1) * 4;
    *out = d;
    memset(&d, 0, sizeof d);
out:
    (void)hipFree(d_flags), (void)hipFree(d_tot);
    free(h_tot), free(h_ent), free(unit_cell), free(blk_unit0);
    if (rc != 0) {
        (void)hipGetLastError();
        spmv_blocked_free(&d);
    }
    return rc;
}

static int g_blk_wgs_per_xcd = 256; /* phase 1: workgroups per XCD (8 per CU: 32 waves, each with 4 x 64 entries in flight) */
extern "C" void spmvk_blocked_set_wgs(int per_xcd)
{
    if (per_xcd >= 1 && per_xcd <= 4096)
        g_blk_wgs_per_xcd = per_xcd;
}

extern "C" int spmvk_blocked(const spmv_dev_blocked *d, const void *x, void *y, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (d->rows <= 0)
        return 0;
    /* phase 1: 8 XCDs x K workgroups; every XCD takes the column blocks b = its id (mod 8) */
    const int k1 = 8 * g_blk_wgs_per_xcd;
    const int g2 = d->nrg < 4 * 4096 ? (d->nrg + 3) / 4 : 4096;
    if (d->elem == 8) {
        hipLaunchKernelGGL((k_blk_sums<double, 4>), dim3(k1), dim3(256), 0, st, (const double *)d->bval, d->bcol, d->unit_cell,
                           d->blk_unit0, d->cell_ent, d->cell_seg, d->nblk, (const double *)x, (double *)d->seg_sum);
        hipLaunchKernelGGL(k_blk_rows<double>, dim3(g2), dim3(256), 0, st, d->cell_mask, d->cell_seg, (const double *)d->seg_sum,
                           (double *)y, d->rows, d->nrg, d->nblk);
    } else {
        hipLaunchKernelGGL((k_blk_sums<float, 4>), dim3(k1), dim3(256), 0, st, (const float *)d->bval, d->bcol, d->unit_cell,
                           d->blk_unit0, d->cell_ent, d->cell_seg, d->nblk, (const float *)x, (float *)d->seg_sum);
        hipLaunchKernelGGL(k_blk_rows<float>, dim3(g2), dim3(256), 0, st, d->cell_mask, d->cell_seg, (const float *)d->seg_sum,
                           (float *)y, d->rows, d->nrg, d->nblk);
    }
    return (int)hipGetLastError();
}
