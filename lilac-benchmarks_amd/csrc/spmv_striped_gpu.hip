/*
 * csrc/spmv_striped_gpu.hip -- the STRIPED conversion on the GPU.
 *
 * spmv_striped.c converts a CSR row block into the STRIPED layout on host threads: three passes over all entries
 * (class C: 0.09-0.16 s of its 0.17-s upload, class D 1.8 of 2.1 s; profiles/r02_upload_breakdown.log), as long as the
 * whole device-resident class C benchmark.  The CSR copy is already in HBM when the layout is wanted, so the passes over
 * the entries run here instead; every DECISION -- stripes and groups, row blocks per group, the row cuts, the tile starts
 * -- is made by the same host functions as in the host builder (spmv_striped_plan_geometry, _alloc_blocks, _cut_all,
 * _tile_starts), from small arrays copied back (one weight per row and group, one size per tile), so both builders
 * produce the SAME layout byte for byte (tests/test_gpu_parity.py compares digests).
 *
 *   k_sg_count    wave per row: entries of the row per stripe, cnt[row][stripe] (rows must be sorted by column: then a
 *                 stripe's entries are one run; an unsorted row, a count beyond 16 bits or too many stripes send the
 *                 caller back to the host builder)
 *   k_sg_weigh    thread per row: wgt[group][row] = sum over the group's stripes of max(1, cnt) (an empty segment costs
 *                 a filler)                                    -> host: block allocation, row cuts
 *   k_sg_offsets  workgroup per wave range: for every stripe of its group, where each row's segment starts inside the
 *                 tile (exclusive scan of max(1, cnt) over the range's rows), the tile's size, the range's fillers
 *                                                              -> host: tile starts
 *   k_sg_scatter  wave per row: every entry to its place (16-bit stripe-local column, row-end flag, value into the
 *                 lane-interleaved slot), a filler for every empty segment
 *   k_sg_pads     thread per tile: the unused tail of every tile = the zero column
 */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "spmv_internal.h"

namespace {

#define SG_MAX_STRIPES 512 /* per-wave counters in LDS */

#define SGCHK(expr)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            spmv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = e_ == hipErrorOutOfMemory ? -3 : -1;                                                \
            goto out;                                                                                \
        }                                                                                            \
    } while (0)

/* flags[0]: some row's columns decrease; flags[1]: a (row, stripe) count beyond 16 bits */
__global__ void __launch_bounds__(256) k_sg_count(const int *__restrict__ rowptr, const int *__restrict__ col, int rows,
                                                  int S, int W, unsigned short *__restrict__ cnt, int *__restrict__ flags)
{
    __shared__ unsigned int s_cnt[4][SG_MAX_STRIPES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bad_order = 0, overflow = 0;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        for (int s = lane; s < S; s += 64)
            s_cnt[w][s] = 0;
        __builtin_amdgcn_wave_barrier();
        const int b = rowptr[r], e = rowptr[r + 1];
        int last = -1; /* column of the entry before this chunk */
        for (int base = b; base < e; base += 64) {
            const int j = base + lane;
            const int c = j < e ? col[j] - 1 : 0x7fffffff;
            int prev = __shfl_up(c, 1, 64);
            if (lane == 0)
                prev = last;
            bad_order |= j < e && c < prev;
            if (j < e)
                atomicAdd(&s_cnt[w][c / W], 1u);
            last = __shfl(c, 63, 64);
        }
        __builtin_amdgcn_wave_barrier();
        for (int s = lane; s < S; s += 64) {
            const unsigned int v = s_cnt[w][s];
            overflow |= v > 65535u;
            cnt[(size_t)r * S + s] = (unsigned short)(v > 65535u ? 65535u : v);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (bad_order)
        flags[0] = 1;
    if (overflow)
        flags[1] = 1;
}

__global__ void __launch_bounds__(256) k_sg_weigh(const unsigned short *__restrict__ cnt, int rows, int S, int NG,
                                                  const int *__restrict__ gs0, unsigned int *__restrict__ wgt)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows)
        return;
    const unsigned short *c = cnt + (size_t)r * S;
    for (int g = 0; g < NG; ++g) {
        unsigned int w = 0;
        for (int s = gs0[g]; s < gs0[g + 1]; ++s)
            w += c[s] ? c[s] : 1u;
        wgt[(size_t)g * rows + r] = w;
    }
}

/* one workgroup per wave range u: rows [rrow[2u], rrow[2u+1]) (at most 1024), stripes [s0, s1) of its group */
__global__ void __launch_bounds__(256) k_sg_offsets(const unsigned short *__restrict__ cnt, int S, int GM, int WPB,
                                                    const int *__restrict__ rrow, const int *__restrict__ wgd,
                                                    unsigned int *__restrict__ off, int *__restrict__ traw,
                                                    int *__restrict__ rfill)
{
    __shared__ unsigned int s_w[4];
    __shared__ unsigned int s_z[4];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r0 = rrow[2 * u], r1 = rrow[2 * u + 1];
    const int s0 = wgd[4 * (u / WPB)], s1 = wgd[4 * (u / WPB) + 1];
    unsigned int zeros_total = 0;
    for (int s = s0; s < s1; ++s) {
        unsigned int v[4], sum = 0, zeros = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + tid * 4 + i;
            const unsigned int c = r < r1 ? cnt[(size_t)r * S + s] : 0xffffffffu;
            v[i] = c == 0xffffffffu ? 0u : (c ? c : 1u);
            zeros += c == 0u;
            sum += v[i];
        }
        /* exclusive scan of `sum` over the 256 threads: inside the wave, then over the four waves */
        unsigned int inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_up(inc, o, 64);
            if (lane >= o)
                inc += t;
        }
        for (int o = 32; o; o >>= 1)
            zeros += __shfl_xor(zeros, o, 64);
        __syncthreads(); /* the previous stripe's s_w is no longer read */
        if (lane == 63)
            s_w[w] = inc, s_z[w] = zeros;
        __syncthreads();
        unsigned int base = inc - sum;
        for (int k = 0; k < w; ++k)
            base += s_w[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + tid * 4 + i;
            if (r < r1)
                off[(size_t)r * S + s] = base;
            base += v[i];
        }
        if (tid == 0) {
            traw[(size_t)u * GM + (s - s0)] = (int)(s_w[0] + s_w[1] + s_w[2] + s_w[3]);
            zeros_total += s_z[0] + s_z[1] + s_z[2] + s_z[3];
        }
    }
    if (tid == 0)
        rfill[u] = (int)zeros_total;
}

template <typename T>
__global__ void __launch_bounds__(256) k_sg_scatter(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                    const T *__restrict__ val, int rows, int S, int W, int NG, int GM,
                                                    int zero_col, const unsigned short *__restrict__ cnt,
                                                    const unsigned int *__restrict__ off, const int *__restrict__ gs0,
                                                    const int *__restrict__ ubase, const int *__restrict__ rrow,
                                                    const int *__restrict__ tptr, unsigned short *__restrict__ tcol,
                                                    T *__restrict__ tval)
{
    constexpr int G = 16 / (int)sizeof(T), EPL = SPMV_STRIPE_EPL, SUB = 64 * EPL;
    __shared__ unsigned int s_first[4][SG_MAX_STRIPES + 1]; /* entries of the row before stripe s */
    __shared__ unsigned int s_tb[4][SG_MAX_STRIPES];        /* where the row's segment of stripe s starts in the stream */
    __shared__ int s_ug[4][64];                             /* the row's wave range in each group */
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const unsigned short *c = cnt + (size_t)r * S;
        /* wave range of row r in every group: the group's ranges are consecutive, their first rows ascend */
        for (int g = lane; g < NG; g += 64) {
            int lo = ubase[g], hi = ubase[g + 1] - 1; /* last range whose first row is <= r */
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (rrow[2 * mid] <= r)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            s_ug[w][g] = lo;
        }
        if (lane == 0) {
            unsigned int a = 0;
            for (int s = 0; s < S; ++s) {
                s_first[w][s] = a;
                a += c[s];
            }
            s_first[w][S] = a;
        }
        __builtin_amdgcn_wave_barrier();
        for (int s = lane; s < S; s += 64) {
            int g = (int)((long long)s * NG / S); /* about right; gs0[g] = floor(g S / NG) <= s < gs0[g + 1] decides */
            while (g > 0 && gs0[g] > s)
                --g;
            while (g + 1 < NG && gs0[g + 1] <= s)
                ++g;
            const int u = s_ug[w][g];
            const unsigned int tb = (unsigned int)tptr[(size_t)u * (GM + 1) + (s - gs0[g])] + off[(size_t)r * S + s];
            s_tb[w][s] = tb;
            if (c[s] == 0) /* the filler of an empty segment: flag only, value 0, times the zero column */
                tcol[tb] = (unsigned short)(0x8000u | (unsigned int)zero_col);
        }
        __builtin_amdgcn_wave_barrier();
        const int b = rowptr[r], e = rowptr[r + 1];
        for (int j = b + lane; j < e; j += 64) {
            const int c0 = col[j] - 1;
            const int s = c0 / W;
            const unsigned int idx = (unsigned int)(j - b) - s_first[w][s];
            const unsigned int n = s_first[w][s + 1] - s_first[w][s];
            const unsigned int p = s_tb[w][s] + idx;
            tcol[p] = (unsigned short)((unsigned int)(c0 - s * W) | (idx + 1 == n ? 0x8000u : 0u));
            const unsigned int blk = p / SUB * SUB, rr = p % SUB, l = rr / EPL, i = rr % EPL;
            tval[blk + (i / G) * (64 * G) + l * G + (i % G)] = val[j];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

/* the rest of every tile is padding: value 0 (the buffer was cleared) times the zero column, no flag */
__global__ void __launch_bounds__(256) k_sg_pads(const int *__restrict__ tptr, const int *__restrict__ traw,
                                                 const int *__restrict__ wgd, int n_ranges, int GM, int WPB, int zero_col,
                                                 unsigned short *__restrict__ tcol)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n_ranges * GM)
        return;
    const int u = (int)(t / GM), k = (int)(t % GM);
    if (k >= wgd[4 * (u / WPB) + 1] - wgd[4 * (u / WPB)])
        return;
    const int b = tptr[(size_t)u * (GM + 1) + k] + traw[(size_t)u * GM + k], e = tptr[(size_t)u * (GM + 1) + k + 1];
    for (int p = b; p < e; ++p)
        tcol[p] = (unsigned short)zero_col;
}

__global__ void __launch_bounds__(256) k_sg_fill16(unsigned short *__restrict__ p, size_t n, int v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = (unsigned short)v;
}

} // namespace

/*
 * Builds the STRIPED layout of the device-resident CSR block (d_rowptr 0-based and local, d_col 1-based, d_val) on the
 * device.  On success (0) t holds the geometry and HOST copies of the small arrays (tptr, rrow, wgd; t->val and t->col
 * stay NULL) and *d_tcol / *d_tval the packed arrays in device memory (n_packed + SPMV_TILE_PAD entries).  -2: not
 * representable / not worth it (the host builder's criteria); -3: out of memory; -4: not a case for this builder (rows
 * not sorted by column, a segment beyond 16 bits, too many stripes): the caller runs the host builder; -1: a HIP error.
 */
extern "C" int spmv_striped_build_device(const int *d_rowptr, const int *d_col, const void *d_val, int rows, int cols,
                                         int elem_size, int64_t nnz, const spmv_striped_opts *opts, void *stream,
                                         spmv_striped_host *t, void **d_tcol_out, void **d_tval_out)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    spmv_striped_plan_geometry(rows, cols, elem_size, nnz, opts, t);
    const int S = t->S, NG = t->NG, GM = t->GM, WPB = t->WPB, W = t->W;
    if (rows <= 0 || opts->wave_rows < 64 || opts->wave_rows % 64 || opts->wave_rows > 1024 || S > 32768 ||
        (int64_t)NG * rows > ((int64_t)1 << 28))
        return -2;
    if (S > SG_MAX_STRIPES || NG > 64 || nnz <= 0)
        return -4;
    if ((int64_t)rows * S * 6 > ((int64_t)16 << 30)) /* the per-(row, stripe) temporaries would not be small beside the matrix */
        return -4;
    unsigned short *d_cnt = nullptr, *d_tcol = nullptr;
    unsigned int *d_off = nullptr, *d_wgt = nullptr;
    int *d_flags = nullptr, *d_gs0 = nullptr, *d_ubase = nullptr, *d_rrow = nullptr, *d_wgd = nullptr, *d_tptr = nullptr;
    int *d_traw = nullptr, *d_rfill = nullptr;
    void *d_tval = nullptr;
    uint32_t *h_wgt = nullptr;
    int *gs0 = nullptr, *gnb = nullptr, *ubase = nullptr, *h_traw = nullptr, *h_rfill = nullptr;
    int64_t *gent = nullptr, *tsize = nullptr, *rfill = nullptr;
    int flags[2] = {0, 0};
    const size_t rs = (size_t)rows * (size_t)S;
    const int grid_rows = rows < 4 * 4096 ? (rows + 3) / 4 : 4096; /* waves stride over the rows */

    gs0 = (int *)malloc(((size_t)NG + 1) * sizeof(int));
    gnb = (int *)malloc((size_t)NG * sizeof(int));
    ubase = (int *)malloc(((size_t)NG + 1) * sizeof(int));
    gent = (int64_t *)calloc((size_t)NG, sizeof(int64_t));
    h_wgt = (uint32_t *)malloc((size_t)NG * rows * sizeof(uint32_t));
    if (!gs0 || !gnb || !ubase || !gent || !h_wgt) {
        rc = -3;
        goto out;
    }
    for (int g = 0; g <= NG; ++g)
        gs0[g] = (int)((int64_t)g * S / NG);

    /* ---- pass 1: entries per (row, stripe) ---- */
    SGCHK(hipMalloc((void **)&d_cnt, rs * sizeof(unsigned short)));
    SGCHK(hipMalloc((void **)&d_flags, 2 * sizeof(int)));
    SGCHK(hipMalloc((void **)&d_gs0, ((size_t)NG + 1) * sizeof(int)));
    SGCHK(hipMalloc((void **)&d_wgt, (size_t)NG * rows * sizeof(unsigned int)));
    SGCHK(hipMemcpyAsync(d_flags, flags, sizeof flags, hipMemcpyHostToDevice, st));
    SGCHK(hipMemcpyAsync(d_gs0, gs0, ((size_t)NG + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_sg_count, dim3(grid_rows), dim3(256), 0, st, d_rowptr, d_col, rows, S, W, d_cnt, d_flags);
    hipLaunchKernelGGL(k_sg_weigh, dim3((rows + 255) / 256), dim3(256), 0, st, d_cnt, rows, S, NG, d_gs0, d_wgt);
    SGCHK(hipGetLastError());
    SGCHK(hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, st));
    SGCHK(hipMemcpyAsync(h_wgt, d_wgt, (size_t)NG * rows * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    SGCHK(hipStreamSynchronize(st));
    if (flags[0] || flags[1]) {
        rc = -4;
        goto out;
    }

    /* ---- the decisions, exactly as the host builder makes them ---- */
    for (int g = 0; g < NG; ++g) {
        const uint32_t *w = h_wgt + (size_t)g * rows;
        int64_t e = 0;
        for (int r = 0; r < rows; ++r)
            e += w[r];
        gent[g] = e;
    }
    spmv_striped_alloc_blocks(opts, rows, gent, gnb, t);
    {
        const int nr = t->n_ranges;
        if ((int64_t)nr * GM > ((int64_t)1 << 26)) {
            rc = -2;
            goto out;
        }
        t->rrow = (int32_t *)calloc((size_t)nr * 2, sizeof(int32_t));
        t->wgd = (int32_t *)calloc((size_t)t->n_wg * 4, sizeof(int32_t));
        t->tptr = (int32_t *)calloc((size_t)nr * ((size_t)GM + 1), sizeof(int32_t));
        tsize = (int64_t *)malloc((size_t)nr * GM * sizeof(int64_t));
        rfill = (int64_t *)malloc((size_t)nr * sizeof(int64_t));
        h_traw = (int *)malloc((size_t)nr * GM * sizeof(int));
        h_rfill = (int *)malloc((size_t)nr * sizeof(int));
        if (!t->rrow || !t->wgd || !t->tptr || !tsize || !rfill || !h_traw || !h_rfill) {
            rc = -3;
            goto out;
        }
        spmv_striped_cut_all(opts, rows, h_wgt, gnb, t);
        ubase[0] = 0;
        for (int g = 0; g < NG; ++g)
            ubase[g + 1] = ubase[g] + gnb[g] * WPB;

        /* ---- pass 2: segment offsets inside the tiles, tile sizes ---- */
        SGCHK(hipFree(d_wgt));
        d_wgt = nullptr;
        SGCHK(hipMalloc((void **)&d_off, rs * sizeof(unsigned int)));
        SGCHK(hipMalloc((void **)&d_rrow, (size_t)nr * 2 * sizeof(int)));
        SGCHK(hipMalloc((void **)&d_wgd, (size_t)t->n_wg * 4 * sizeof(int)));
        SGCHK(hipMalloc((void **)&d_traw, (size_t)nr * GM * sizeof(int)));
        SGCHK(hipMalloc((void **)&d_rfill, (size_t)nr * sizeof(int)));
        SGCHK(hipMalloc((void **)&d_ubase, ((size_t)NG + 1) * sizeof(int)));
        SGCHK(hipMemcpyAsync(d_rrow, t->rrow, (size_t)nr * 2 * sizeof(int), hipMemcpyHostToDevice, st));
        SGCHK(hipMemcpyAsync(d_wgd, t->wgd, (size_t)t->n_wg * 4 * sizeof(int), hipMemcpyHostToDevice, st));
        SGCHK(hipMemcpyAsync(d_ubase, ubase, ((size_t)NG + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        SGCHK(hipMemsetAsync(d_traw, 0, (size_t)nr * GM * sizeof(int), st));
        hipLaunchKernelGGL(k_sg_offsets, dim3(nr), dim3(256), 0, st, d_cnt, S, GM, WPB, d_rrow, d_wgd, d_off, d_traw, d_rfill);
        SGCHK(hipGetLastError());
        SGCHK(hipMemcpyAsync(h_traw, d_traw, (size_t)nr * GM * sizeof(int), hipMemcpyDeviceToHost, st));
        SGCHK(hipMemcpyAsync(h_rfill, d_rfill, (size_t)nr * sizeof(int), hipMemcpyDeviceToHost, st));
        SGCHK(hipStreamSynchronize(st));
        {
            const int64_t sub = 64 * (int64_t)t->EPL;
            for (size_t i = 0; i < (size_t)nr * GM; ++i)
                tsize[i] = ((int64_t)h_traw[i] + sub - 1) / sub * sub;
            for (int u = 0; u < nr; ++u)
                rfill[u] = h_rfill[u];
        }
        if (spmv_striped_tile_starts(tsize, rfill, nnz, t) != 0) {
            rc = -2;
            goto out;
        }

        /* ---- pass 3: the entries to their places ---- */
        const size_t n_alloc = (size_t)t->n_packed + SPMV_TILE_PAD;
        SGCHK(hipMalloc((void **)&d_tcol, n_alloc * sizeof(unsigned short)));
        SGCHK(hipMalloc(&d_tval, n_alloc * (size_t)elem_size));
        SGCHK(hipMalloc((void **)&d_tptr, (size_t)nr * ((size_t)GM + 1) * sizeof(int)));
        SGCHK(hipMemcpyAsync(d_tptr, t->tptr, (size_t)nr * ((size_t)GM + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        SGCHK(hipMemsetAsync(d_tval, 0, n_alloc * (size_t)elem_size, st));
        hipLaunchKernelGGL(k_sg_fill16, dim3(16), dim3(256), 0, st, d_tcol + t->n_packed, (size_t)SPMV_TILE_PAD, t->cap - 1);
        if (elem_size == 8)
            hipLaunchKernelGGL(k_sg_scatter<double>, dim3(grid_rows), dim3(256), 0, st, d_rowptr, d_col, (const double *)d_val,
                               rows, S, W, NG, GM, t->cap - 1, d_cnt, d_off, d_gs0, d_ubase, d_rrow, d_tptr, d_tcol,
                               (double *)d_tval);
        else
            hipLaunchKernelGGL(k_sg_scatter<float>, dim3(grid_rows), dim3(256), 0, st, d_rowptr, d_col, (const float *)d_val,
                               rows, S, W, NG, GM, t->cap - 1, d_cnt, d_off, d_gs0, d_ubase, d_rrow, d_tptr, d_tcol,
                               (float *)d_tval);
        hipLaunchKernelGGL(k_sg_pads, dim3((unsigned int)(((size_t)nr * GM + 255) / 256)), dim3(256), 0, st, d_tptr, d_traw, d_wgd,
                           nr, GM, WPB, t->cap - 1, d_tcol);
        SGCHK(hipGetLastError());
        SGCHK(hipStreamSynchronize(st));
    }
    *d_tcol_out = d_tcol;
    *d_tval_out = d_tval;
    d_tcol = nullptr;
    d_tval = nullptr;
out:
    (void)hipFree(d_cnt), (void)hipFree(d_off), (void)hipFree(d_wgt), (void)hipFree(d_flags), (void)hipFree(d_gs0);
    (void)hipFree(d_ubase), (void)hipFree(d_rrow), (void)hipFree(d_wgd), (void)hipFree(d_tptr), (void)hipFree(d_traw);
    (void)hipFree(d_rfill), (void)hipFree(d_tcol), (void)hipFree(d_tval);
    free(gs0), free(gnb), free(ubase), free(gent), free(h_wgt), free(tsize), free(rfill), free(h_traw), free(h_rfill);
    if (rc != 0) {
        (void)hipGetLastError();
        spmv_striped_free(t);
    }
    return rc;
}
