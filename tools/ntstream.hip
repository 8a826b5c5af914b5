// tools/ntstream.hip -- what can a kernel that only READS reach on this chip, with the load shapes and the launch shape of
// k_striped (256 workgroups x 16 waves, every wave walking a private linear stream in 256-entry sub-windows: 8 bytes of
// 16-bit columns + two dense 16-byte value loads per lane, a ring of two register sets), with ordinary and with
// non-temporal loads, at the sizes of class C's packed stream, of a class D row block (1/8) and of all of class D?
// The answer replaces round 1's "measured read ceiling" (6.39 TB/s, ordinary loads, 512 MB), which the product has since
// overtaken on class D.
//   hipcc -O3 --offload-arch=gfx950 tools/ntstream.hip -o tools/bin/ntstream
//   tools/bin/ntstream [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename V, bool NT> __device__ __forceinline__ V ld(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}

// wave g walks sub-windows [g * steps, (g + 1) * steps); a sub-window = 256 entries: 512 B of columns, 2 KiB of values
template <int R, bool NT> __global__ void __launch_bounds__(1024) k_wave_streams(const double* __restrict__ val, const unsigned short* __restrict__ col,
                                                                                 size_t steps, float* sink) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 16 + w;
    d2 va[R], vb[R]; u2 cc[R];
    auto load = [&](int d, size_t t) {
        const size_t base = (g * steps + t) * 256;
        cc[d] = ld<u2, NT>((const u2*)(col + base + 4 * lane));
        va[d] = ld<d2, NT>((const d2*)(val + base + 2 * lane));
        vb[d] = ld<d2, NT>((const d2*)(val + base + 128 + 2 * lane));
    };
#pragma unroll
    for (int d = 0; d < R; ++d) load(d, d);
    double acc = 0; unsigned int ca = 0;
    for (size_t t = 0; t + R <= steps; t += R) {
#pragma unroll
        for (int d = 0; d < R; ++d) {
            acc += va[d][0] + va[d][1] + vb[d][0] + vb[d][1]; ca += cc[d][0] ^ cc[d][1];
            const size_t nt = t + d + R;
            load(d, nt < steps ? nt : d);
        }
    }
    if (acc == 12345.678 && ca == 77) *sink = 1;
}

// grid-stride 16-byte reads, U in flight per lane: the plainest reading kernel
template <int U, bool NT> __global__ void __launch_bounds__(256) k_seq(const f4* __restrict__ p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<f4, NT>(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = 1;
}

template <typename F> float time_us(F f, int reps) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    float* sink; CHECK(hipMalloc(&sink, 4));
    struct { const char* name; size_t entries; } cases[] = {
        {"class C packed stream (36.56 M entries)", 36560000}, {"class D row block 1/8 (86.8 M)", 86800000}, {"all of class D (694 M)", 694400000}};
    printf("%-44s %-34s %10s %10s\n", "stream", "kernel", "us", "GB/s");
    for (auto& c : cases) {
        const size_t waves = 256 * 16;
        const size_t steps = c.entries / 256 / waves / 2 * 2;  // whole ring turns
        const size_t entries = steps * waves * 256, bytes = entries * 10;
        double* val; unsigned short* col;
        CHECK(hipMalloc(&val, entries * 8 + (1 << 20))); CHECK(hipMalloc(&col, entries * 2 + (1 << 20)));
        CHECK(hipMemset(val, 0, entries * 8 + (1 << 20))); CHECK(hipMemset(col, 0, entries * 2 + (1 << 20)));
        const int r = c.entries > 400000000 ? (reps + 5) / 6 : reps;
        for (int round = 0; round < 2; ++round) {  // interleaved A/B, twice
            float t;
            t = time_us([&] { k_wave_streams<2, false><<<256, 1024>>>(val, col, steps, sink); }, r);
            printf("%-44s %-34s %10.1f %10.1f\n", c.name, "16-wave streams R=2, ordinary loads", t, bytes / t / 1e3);
            t = time_us([&] { k_wave_streams<2, true><<<256, 1024>>>(val, col, steps, sink); }, r);
            printf("%-44s %-34s %10.1f %10.1f\n", c.name, "16-wave streams R=2, non-temporal", t, bytes / t / 1e3);
            t = time_us([&] { k_wave_streams<4, true><<<256, 1024>>>(val, col, steps, sink); }, r);
            printf("%-44s %-34s %10.1f %10.1f\n", c.name, "16-wave streams R=4, non-temporal", t, bytes / t / 1e3);
        }
        const size_t n = entries * 8 / 16;
        float t = time_us([&] { k_seq<8, false><<<2048, 256>>>((const f4*)val, n, sink); }, r);
        printf("%-44s %-34s %10.1f %10.1f\n", c.name, "grid-stride 16 B x 8, ordinary", t, n * 16 / t / 1e3);
        t = time_us([&] { k_seq<8, true><<<2048, 256>>>((const f4*)val, n, sink); }, r);
        printf("%-44s %-34s %10.1f %10.1f\n", c.name, "grid-stride 16 B x 8, non-temporal", t, n * 16 / t / 1e3);
        t = time_us([&] { k_seq<8, true><<<4096, 256>>>((const f4*)val, n, sink); }, r);
        printf("%-44s %-34s %10.1f %10.1f\n", c.name, "grid-stride x 8, NT, 4096 workgroups", t, n * 16 / t / 1e3);
        fflush(stdout);
        CHECK(hipFree(val)); CHECK(hipFree(col));
    }
    return 0;
}
