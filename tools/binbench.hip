// tools/binbench.hip -- feasibility of a ROW-BINNED two-phase product for matrices whose x is far larger than a cache that
// gathers well and whose rows are too short for STRIPED (SparseBench 170^3: x = 39 MB, 4.9 entries per row; STREAM runs it
// at the chip's gather rate, 370 us = 0.13 of 8 TB/s).
//   phase 1: entries in (column stripe, row bin, row) order; a workgroup keeps ITS stripe of x in LDS (up to 160 KB), streams
//            (value, 16-bit column), multiplies, and writes the products to their place in phase 2's order: both orders keep a
//            (stripe, bin) cell contiguous and in the same inner order, so one 4-byte destination per four products is enough;
//   phase 2: products in (row bin, column stripe, row) order next to a 16-bit bin-local row; one wave per bin adds them into
//            its slice of y in LDS and writes the slice.
// Everything is a linear stream except the 32-byte product chunks of phase 1 (cells of ~35 entries: ~300-byte runs).
// Synthetic matrix of SparseBench's shape generated here; both phases timed; y checked against a host CSR product.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/binbench.hip -o /tmp/binbench && /tmp/binbench [size] [bin_rows] [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

constexpr int WIN = 256;        // entries per window: 64 lanes x 4
constexpr int XCAP = 19456;     // doubles of x per stripe in LDS (152 KB), the last one is the zero column
constexpr int P1_WAVES = 16;

// workgroup q: stripe wg[3q], windows [wg[3q+1], wg[3q+2]) of the phase-1 stream
template <int U>
__global__ void __launch_bounds__(P1_WAVES * 64) k_products(const int *__restrict__ wg, const double *__restrict__ val,
                                                            const unsigned short *__restrict__ col,
                                                            const unsigned int *__restrict__ dest4, const double *__restrict__ x,
                                                            int cols, int wcols, double *__restrict__ prod)
{
    __shared__ __attribute__((aligned(16))) double s_x[XCAP];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int s = wg[3 * blockIdx.x], w0 = wg[3 * blockIdx.x + 1], w1 = wg[3 * blockIdx.x + 2];
    const int c0 = s * wcols;
    const int nx = min(wcols, cols - c0);
    for (int i = tid * 2; i < XCAP; i += P1_WAVES * 64 * 2) {
        d2 v;
        v[0] = i < nx ? x[c0 + i] : 0.0;
        v[1] = i + 1 < nx ? x[c0 + i + 1] : 0.0;
        *reinterpret_cast<d2 *>(&s_x[i]) = v;
    }
    __syncthreads();
    const int n = w1 - w0;
    const int a = w0 + (int)((long long)n * w / P1_WAVES), b = w0 + (int)((long long)n * (w + 1) / P1_WAVES);
    if (a >= b)
        return;
    d2 v[U][2];
    u2 c[U];
    unsigned int d[U];
    auto load = [&](int u, int kk) __attribute__((always_inline)) {
        kk = kk < b ? kk : b - 1;
        const size_t base = (size_t)kk * WIN;
        v[u][0] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(val + base) + lane);
        v[u][1] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(val + base + 128) + lane);
        c[u] = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(col + base) + lane);
        d[u] = __builtin_nontemporal_load(dest4 + base / 4 + lane);
    };
#pragma unroll
    for (int u = 0; u < U; ++u)
        load(u, a + u);
    for (int k = a; k < b; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d2 p0, p1;
            p0[0] = v[u][0][0] * s_x[c[u][0] & 0xffffu];
            p0[1] = v[u][0][1] * s_x[c[u][0] >> 16];
            p1[0] = v[u][1][0] * s_x[c[u][1] & 0xffffu];
            p1[1] = v[u][1][1] * s_x[c[u][1] >> 16];
            d2 *out = reinterpret_cast<d2 *>(prod + (size_t)d[u] * 4);
            load(u, k + u + U);
            if (k + u < b) {
                out[0] = p0;
                out[1] = p1;
            }
        }
    }
}

// one wave per bin: windows [bw[b], bw[b+1]) of the phase-2 stream, rows [b*rb, ...)
template <int U, int RB, int MODE, bool NTL>
__global__ void __launch_bounds__(256) k_bins(const int *__restrict__ bw, const double *__restrict__ prod,
                                              const unsigned short *__restrict__ lrow, int rows, int nbins,
                                              double *__restrict__ y)
{
    __shared__ __attribute__((aligned(16))) double s_y[4][RB];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + w;
    if (b >= nbins)
        return;
    for (int i = lane; i < RB; i += 64)
        s_y[w][i] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int a = bw[b], e = bw[b + 1];
    if (a < e) {
        d2 v[U][2];
        unsigned int r[U][2];
        auto load = [&](int u, int kk) __attribute__((always_inline)) {
            kk = kk < e ? kk : e - 1;
            const size_t base = (size_t)kk * WIN;
            if (NTL) {
                v[u][0] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(prod + base) + lane);
                v[u][1] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(prod + base + 128) + lane);
            } else {
                v[u][0] = reinterpret_cast<const d2 *>(prod + base)[lane];
                v[u][1] = reinterpret_cast<const d2 *>(prod + base + 128)[lane];
            }
            r[u][0] = __builtin_nontemporal_load(reinterpret_cast<const unsigned int *>(lrow + base) + lane);
            r[u][1] = __builtin_nontemporal_load(reinterpret_cast<const unsigned int *>(lrow + base + 128) + lane);
        };
#pragma unroll
        for (int u = 0; u < U; ++u)
            load(u, a + u);
        for (int k = a; k < e; k += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const d2 q0 = v[u][0], q1 = v[u][1];
                const unsigned int r0 = r[u][0], r1 = r[u][1];
                load(u, k + u + U);
                if (k + u < e) {
                    if (MODE == 0) {
                        unsafeAtomicAdd(&s_y[w][r0 & 0xffffu], q0[0]);
                        unsafeAtomicAdd(&s_y[w][r0 >> 16], q0[1]);
                        unsafeAtomicAdd(&s_y[w][r1 & 0xffffu], q1[0]);
                        unsafeAtomicAdd(&s_y[w][r1 >> 16], q1[1]);
                    } else if (MODE == 1) { /* timing only: racy */
                        s_y[w][r0 & 0xffffu] += q0[0];
                        s_y[w][r0 >> 16] += q0[1];
                        s_y[w][r1 & 0xffffu] += q1[0];
                        s_y[w][r1 >> 16] += q1[1];
                    } else { /* timing only: loads alone */
                        s_y[w][lane] += q0[0] + q0[1] + q1[0] + q1[1] + (double)(r0 ^ r1);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int r0 = b * RB;
    for (int i = lane; i < RB && r0 + i < rows; i += 64)
        y[r0 + i] = s_y[w][i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int g_mode = 0, g_ntl = 1;
template <int RB> static void launch_bins(int nb, const int *bw, const double *prod, const unsigned short *lrow, int rows, double *y)
{
#define LB(M, N) hipLaunchKernelGGL((k_bins<4, RB, M, N>), dim3((nb + 3) / 4), dim3(256), 0, 0, bw, prod, lrow, rows, nb, y)
    if (g_mode == 0 && g_ntl) LB(0, true);
    else if (g_mode == 0) LB(0, false);
    else if (g_mode == 1) LB(1, true);
    else LB(2, true);
}

int main(int argc, char **argv)
{
    const int size = argc > 1 ? atoi(argv[1]) : 170;
    const int RB = argc > 2 ? atoi(argv[2]) : 2048;
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    g_mode = argc > 4 ? atoi(argv[4]) : 0;
    g_ntl = argc > 5 ? atoi(argv[5]) : 1;
    const int n = size * size * size;
    double t0 = now();
    // ---- SparseBench-shaped CSR: max(1, trunc(N(5, 2))) distinct uniform columns per row, sorted
    std::vector<int> rowptr(n + 1);
    std::vector<int> col;
    std::vector<double> val;
    {
        std::mt19937_64 g(170);
        std::normal_distribution<double> nd(5.0, 2.0), vd(0.0, 2.0);
        col.reserve((size_t)n * 5 + 1024);
        std::vector<int> tmp;
        rowptr[0] = 0;
        for (int r = 0; r < n; ++r) {
            int k = (int)nd(g);
            if (k < 1) k = 1;
            tmp.clear();
            while ((int)tmp.size() < k) {
                int c = (int)(g() % (uint64_t)n);
                if (std::find(tmp.begin(), tmp.end(), c) == tmp.end()) tmp.push_back(c);
            }
            std::sort(tmp.begin(), tmp.end());
            for (int c : tmp) col.push_back(c);
            rowptr[r + 1] = (int)col.size();
        }
        val.resize(col.size());
        for (auto &v : val) v = std::fabs(vd(g));
    }
    const int64_t nnz = (int64_t)col.size();
    printf("n=%d nnz=%lld generated in %.1f s\n", n, (long long)nnz, now() - t0);
    t0 = now();
    // ---- geometry
    int k = 1;
    while ((int64_t)256 * k * (XCAP - 2) < n) ++k;
    const int S = 256 * k;
    int wcols = (n + S - 1) / S;
    wcols += wcols & 1;
    const int B = (n + RB - 1) / RB;
    printf("S=%d stripes of %d columns (%.1f KB), B=%d bins of %d rows, %.1f entries per cell\n", S, wcols, wcols * 8 / 1024.0, B, RB,
           (double)nnz / ((double)S * B));
    // ---- cells
    std::vector<int> cnt((size_t)S * B, 0);
    for (int r = 0; r < n; ++r)
        for (int j = rowptr[r]; j < rowptr[r + 1]; ++j)
            ++cnt[(size_t)(col[j] / wcols) * B + r / RB];
    std::vector<int64_t> c1((size_t)S * B), c2((size_t)S * B); // c2 indexed [s*B+b] too, holding the phase-2 start of cell (b, s)
    std::vector<int> pad((size_t)S * B);
    std::vector<int64_t> stripe_w0(S + 1);
    int64_t p = 0;
    for (int s = 0; s < S; ++s) {
        stripe_w0[s] = p / WIN;
        int lastb = -1;
        for (int b = 0; b < B; ++b) {
            const size_t id = (size_t)s * B + b;
            pad[id] = (cnt[id] + 3) & ~3;
            c1[id] = p;
            p += pad[id];
            if (cnt[id]) lastb = b;
        }
        const int64_t q = (p + WIN - 1) / WIN * WIN;
        if (q != p && lastb >= 0) { // the stripe's last cell grows to the window boundary (later cells are empty)
            pad[(size_t)s * B + lastb] += (int)(q - p);
            for (int b = lastb + 1; b < B; ++b) c1[(size_t)s * B + b] = q;
        }
        p = q;
    }
    stripe_w0[S] = p / WIN;
    const int64_t n1 = p;
    std::vector<int> bw(B + 1);
    p = 0;
    for (int b = 0; b < B; ++b) {
        bw[b] = (int)(p / WIN);
        for (int s = 0; s < S; ++s) {
            c2[(size_t)s * B + b] = p;
            p += pad[(size_t)s * B + b];
        }
        p = (p + WIN - 1) / WIN * WIN;
    }
    bw[B] = (int)(p / WIN);
    const int64_t n2 = p;
    printf("phase-1 stream %lld entries (%.3f x nnz), phase-2 stream %lld (%.3f x)\n", (long long)n1, (double)n1 / nnz, (long long)n2,
           (double)n2 / nnz);
    std::vector<double> val1(n1, 0.0);
    std::vector<unsigned short> col1(n1, (unsigned short)(XCAP - 1)), lrow(n2, 0);
    std::vector<unsigned int> dest4(n1 / 4);
    {
        std::vector<int> cur((size_t)S * B, 0);
        for (int r = 0; r < n; ++r)
            for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) {
                const int s = col[j] / wcols, b = r / RB;
                const size_t id = (size_t)s * B + b;
                const int kk = cur[id]++;
                const int64_t p1 = c1[id] + kk, p2 = c2[id] + kk;
                const int64_t wb = p1 / WIN * WIN, o = p1 % WIN, l = o / 4, i = o % 4;
                val1[wb + (i / 2) * 128 + l * 2 + (i % 2)] = val[j];
                col1[p1] = (unsigned short)(col[j] - s * wcols);
                const int64_t wb2 = p2 / WIN * WIN, o2 = p2 % WIN; // phase 2 reads entries (2l, 2l+1) and (128+2l, 128+2l+1): linear
                (void)wb2, (void)o2;
                lrow[p2] = (unsigned short)(r - b * RB);
            }
        for (size_t id = 0; id < (size_t)S * B; ++id)
            for (int q = 0; q < pad[id]; q += 4)
                dest4[(c1[id] + q) / 4] = (unsigned int)((c2[id] + q) / 4);
    }
    // phase 1 writes a lane's four products to 4 consecutive phase-2 positions: phase 2 must read them linearly
    // workgroups: one per stripe, heavy stripes split
    std::vector<int> wg;
    {
        const int64_t target = (stripe_w0[S] + S - 1) / S;
        for (int s = 0; s < S; ++s) {
            const int64_t a = stripe_w0[s], e = stripe_w0[s + 1];
            const int parts = (int)std::max<int64_t>(1, (e - a + target * 5 / 4 - 1) / (target * 5 / 4));
            for (int q = 0; q < parts; ++q) {
                wg.push_back(s);
                wg.push_back((int)(a + (e - a) * q / parts));
                wg.push_back((int)(a + (e - a) * (q + 1) / parts));
            }
        }
    }
    const int nwg = (int)wg.size() / 3;
    printf("layout built in %.1f s; %d phase-1 workgroups\n", now() - t0, nwg);

    std::vector<double> x(n), yref(n);
    {
        std::mt19937_64 g(1);
        std::normal_distribution<double> nd(0.0, 1.0);
        for (auto &v : x) v = nd(g);
        for (int r = 0; r < n; ++r) {
            double s = 0.0;
            for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) s += val[j] * x[col[j]];
            yref[r] = s;
        }
    }
    double *d_val, *d_x, *d_prod, *d_y;
    unsigned short *d_col, *d_lrow;
    unsigned int *d_dest;
    int *d_wg, *d_bw;
    CHECK(hipMalloc(&d_val, n1 * 8));
    CHECK(hipMalloc(&d_col, n1 * 2));
    CHECK(hipMalloc(&d_dest, n1));
    CHECK(hipMalloc(&d_prod, n2 * 8));
    CHECK(hipMalloc(&d_lrow, n2 * 2));
    CHECK(hipMalloc(&d_x, (size_t)n * 8));
    CHECK(hipMalloc(&d_y, (size_t)n * 8));
    CHECK(hipMalloc(&d_wg, wg.size() * 4));
    CHECK(hipMalloc(&d_bw, bw.size() * 4));
    CHECK(hipMemcpy(d_val, val1.data(), n1 * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_col, col1.data(), n1 * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_dest, dest4.data(), n1, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_lrow, lrow.data(), n2 * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_x, x.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_wg, wg.data(), wg.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_bw, bw.data(), bw.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_prod, 0, n2 * 8));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventCreate(&e2));
    auto phase1 = [&]() { hipLaunchKernelGGL((k_products<4>), dim3(nwg), dim3(P1_WAVES * 64), 0, 0, d_wg, d_val, d_col, d_dest, d_x, n, wcols, d_prod); };
    auto phase2 = [&]() {
        if (RB == 1024) launch_bins<1024>(B, d_bw, d_prod, d_lrow, n, d_y);
        else if (RB == 2048) launch_bins<2048>(B, d_bw, d_prod, d_lrow, n, d_y);
        else if (RB == 4096) launch_bins<4096>(B, d_bw, d_prod, d_lrow, n, d_y);
        else { printf("bin_rows must be 1024, 2048 or 4096\n"); exit(1); }
    };
    for (int i = 0; i < 5; ++i) phase1(), phase2();
    CHECK(hipDeviceSynchronize());
    float t1 = 0, t2 = 0, tb = 0;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0));
        phase1();
        CHECK(hipEventRecord(e1));
        phase2();
        CHECK(hipEventRecord(e2));
        CHECK(hipEventSynchronize(e2));
        float a, b;
        CHECK(hipEventElapsedTime(&a, e0, e1));
        CHECK(hipEventElapsedTime(&b, e1, e2));
        t1 += a, t2 += b;
    }
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) phase1(), phase2();
    CHECK(hipEventRecord(e2));
    CHECK(hipEventSynchronize(e2));
    CHECK(hipEventElapsedTime(&tb, e0, e2));
    std::vector<double> y(n);
    CHECK(hipMemcpy(y.data(), d_y, (size_t)n * 8, hipMemcpyDeviceToHost));
    double maxrel = 0;
    for (int r = 0; r < n; ++r) {
        double ab = 0;
        for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) ab += std::fabs(val[j] * x[col[j]]);
        const double e = std::fabs(y[r] - yref[r]) / (ab > 0 ? ab : 1);
        if (e > maxrel) maxrel = e;
    }
    const double alg = (double)nnz * 12 + (n + 1) * 4.0 + n * 16.0;
    printf("phase 1 %.1f us, phase 2 %.1f us (events around each); back to back %.1f us per product = %.3f of 8 TB/s on %.0f MB "
           "algorithmic; max error / sum|a x| = %.2e\n", t1 / reps * 1e3, t2 / reps * 1e3, tb / reps * 1e3,
           alg / (tb / reps * 1e-3) / 8e12, alg / 1e6, maxrel);
    return maxrel < 1e-14 || g_mode ? 0 : 2;
}
