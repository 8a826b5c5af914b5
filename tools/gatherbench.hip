// tools/gatherbench.hip -- how fast can a CU gather 8-byte (or 4-byte) words from a table that
// sits in L2 / in the Infinity Cache / in HBM, and does the cache policy of the load matter?
// Index stream read coalesced (16 B per lane), table[idx] gathered, sums written out.
//   hipcc --offload-arch=gfx950 -O3 tools/gatherbench.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int i4 __attribute__((ext_vector_type(4)));

enum { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3, NTSC = 4 };

template <int MODE> __device__ __forceinline__ double gather(const double* p)
{
    double v;
    if constexpr (MODE == PLAIN) {
        v = *p;
    } else if constexpr (MODE == NT) {
        v = __builtin_nontemporal_load(p);
    } else if constexpr (MODE == SC1) {
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if constexpr (MODE == SC0SC1) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else {
        asm volatile("global_load_dwordx2 %0, %1, off nt sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    }
    return v;
}

// U independent 16-byte index loads per thread -> 4*U gathers in flight per lane
template <int MODE, int U>
__global__ void __launch_bounds__(256) k_gather(const int* __restrict__ idx, const double* __restrict__ tab,
                                                double* __restrict__ out, size_t n4)
{
    double acc = 0;
    for (size_t j = (size_t)blockIdx.x * 256 * U + threadIdx.x; j < n4; j += (size_t)gridDim.x * 256 * U) {
        i4 c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t jj = j + (size_t)u * 256;
            c[u] = reinterpret_cast<const i4*>(idx)[jj < n4 ? jj : j];
        }
        if constexpr (MODE == PLAIN || MODE == NT) {
            double g[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    g[u][i] = gather<MODE>(tab + c[u][i]);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc += g[u][i];
        } else {
            // inline-asm flavours: issue all, one wait at the end (hand-written, no per-load wait)
            double g[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double* p = tab + c[u][i];
                    if constexpr (MODE == SC1)
                        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(g[u][i]) : "v"(p) : "memory");
                    else if constexpr (MODE == SC0SC1)
                        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(g[u][i]) : "v"(p) : "memory");
                    else
                        asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(g[u][i]) : "v"(p) : "memory");
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc += g[u][i];
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i)
        f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const size_t n = (size_t)(argc > 1 ? atol(argv[1]) : 32) << 20; // gathers
    const int grid = 256 * 8;
    int* idx;
    double *tab, *out;
    CHECK(hipMalloc(&idx, n * 4 + 4096));
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
    const size_t tab_max = (size_t)64 << 20; // doubles: 512 MB
    CHECK(hipMalloc(&tab, tab_max * 8));
    CHECK(hipMemset(tab, 0, tab_max * 8));
    std::vector<int> h(n);
    const size_t sizes[] = {(size_t)128 << 10, (size_t)1 << 20, (size_t)5 << 20, (size_t)32 << 20}; // 1, 8, 40, 256 MB
    for (size_t m : sizes) {
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < n; ++i) {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            h[i] = (int)(s % m);
        }
        CHECK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
        printf("== table %zu MB, %zu M gathers of 8 B\n", m * 8 >> 20, n >> 20);
#define RUN(MODE, U)                                                                                             \
    {                                                                                                            \
        float ms = time_ms([&] { k_gather<MODE, U><<<grid, 256>>>(idx, tab, out, n / 4); }, 5);                  \
        printf("  %-8s U=%d  %8.1f us  %7.1f G gathers/s  %5.3f per clk per CU (2.4 GHz)\n", #MODE, U, ms * 1e3, \
               n / ms / 1e6, n / ms / 1e6 / (256 * 2.4));                                                        \
    }
        RUN(PLAIN, 1) RUN(PLAIN, 2) RUN(PLAIN, 4) RUN(NT, 2) RUN(SC1, 2) RUN(SC0SC1, 2) RUN(NTSC, 2)
    }
    return 0;
}
