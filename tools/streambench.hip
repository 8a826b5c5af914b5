// tools/streambench.hip -- read-bandwidth ceilings on MI355X for the access patterns our kernels use.
//   hipcc -O3 --offload-arch=gfx950 tools/streambench.hip -o lilac-benchmarks_amd/lib/streambench
// Patterns (all read `bytes` once, 16 B per lane per load, sum into a sink):
//   seq        grid-stride over the whole buffer, U loads in flight per lane
//   wgchunk    each 256-thread workgroup owns one contiguous 24 KiB chunk (the STREAM kernel's pattern)
//   wavestream each wave owns one contiguous region and walks it in 1 KiB steps, D steps in flight
//              (the TILED kernel's pattern: thousands of independent linear streams)
//   wginter    same waves, but the 8 waves of a workgroup interleave their 1 KiB steps in memory
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U> __global__ void __launch_bounds__(256) k_seq(const f4* __restrict__ p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    f4 acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = 1;
}
// each WG: chunk of CH f4 elements (CH = 1536 -> 24 KiB), 6 loads per thread all in flight
__global__ void __launch_bounds__(256) k_wgchunk(const f4* __restrict__ p, size_t n, float* sink) {
    const size_t base = (size_t)blockIdx.x * 1536;
    f4 v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) { size_t i = base + u * 256 + threadIdx.x; v[u] = i < n ? p[i] : f4(0); }
    f4 acc = 0;
#pragma unroll
    for (int u = 0; u < 6; ++u) acc += v[u];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = 1;
}
// wave-private streams: wave g walks [g*len, (g+1)*len) f4 elements in steps of 64 (1 KiB), ring of D
template <int D, bool INTER> __global__ void __launch_bounds__(512, 4) k_wavestream(const f4* __restrict__ p, size_t len, float* sink) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 8 + w;
    const size_t steps = len / 64;
    auto addr = [&](size_t t) -> size_t {
        if (INTER) return ((size_t)blockIdx.x * 8 * steps + t * 8 + w) * 64 + lane;  // waves interleaved per step
        return g * len + t * 64 + lane;
    };
    f4 ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = p[addr(d)];
    f4 acc = 0;
    for (size_t t = 0; t + D <= steps; t += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            acc += ring[d];
            size_t nt = t + d + D;
            ring[d] = p[addr(nt < steps ? nt : d)];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = 1;
}

// Does the Infinity Cache reward a traversal that alternates direction between passes?  Same wave-private
// streams as k_wavestream, but pass `rev` walks every region back to front.
template <int D> __global__ void __launch_bounds__(512, 4) k_wavestream_dir(const f4* __restrict__ p, size_t len, int rev, float* sink) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 8 + w;
    const size_t steps = len / 64;
    auto addr = [&](size_t t) -> size_t { return g * len + (rev ? steps - 1 - t : t) * 64 + lane; };
    f4 ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = p[addr(d)];
    f4 acc = 0;
    for (size_t t = 0; t + D <= steps; t += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            acc += ring[d];
            size_t nt = t + d + D;
            ring[d] = p[addr(nt < steps ? nt : d)];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = 1;
}

// TILED kernel's load shapes: per step a wave takes 256 entries = 2 KiB of f64 values + 512 B of u16 columns.
//   SHAPE 0: lane l owns entries 4l..4l+3: one 8-byte column load + two 16-byte value loads at a 32-byte lane stride
//   SHAPE 1: lane l owns entries {2l,2l+1} and {128+2l,128+2l+1}: two dense 16-byte value loads + two 4-byte column loads
//   SHAPE 2: like 0 but the 8 waves of a workgroup interleave their steps in memory
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
template <int D, int SHAPE> __global__ void __launch_bounds__(512, 4) k_tileshape(const double* __restrict__ val, const unsigned short* __restrict__ col, size_t steps, float* sink) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 8 + w;
    d2 va[D], vb[D]; u2 cc[D];
    auto load = [&](int d, size_t t) {
        size_t base = (SHAPE == 2) ? (((size_t)blockIdx.x * steps + t) * 8 + w) * 256 : (g * steps + t) * 256;
        if (SHAPE == 1) {
            va[d] = *(const d2*)(val + base + 2 * lane);
            vb[d] = *(const d2*)(val + base + 128 + 2 * lane);
            cc[d][0] = *(const unsigned int*)(col + base + 2 * lane);
            cc[d][1] = *(const unsigned int*)(col + base + 128 + 2 * lane);
        } else {
            va[d] = *(const d2*)(val + base + 4 * lane);
            vb[d] = *(const d2*)(val + base + 4 * lane + 2);
            cc[d] = *(const u2*)(col + base + 4 * lane);
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, d);
    double acc = 0; unsigned int ca = 0;
    for (size_t t = 0; t + D <= steps; t += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            acc += va[d][0] + va[d][1] + vb[d][0] + vb[d][1]; ca += cc[d][0] ^ cc[d][1];
            size_t nt = t + d + D;
            load(d, nt < steps ? nt : d);
        }
    }
    if (acc == 12345.678 && ca == 77) *sink = 1;
}

// One interleaved stream instead of two arrays: per step a wave reads ONE contiguous 2560-byte record
// = [256 x u16 columns][256 x f64 values]; lane l takes 8 B of columns and two 16-byte value units.
template <int D> __global__ void __launch_bounds__(512, 4) k_recshape(const char* __restrict__ rec, size_t steps, float* sink) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 8 + w;
    d2 va[D], vb[D]; u2 cc[D];
    auto load = [&](int d, size_t t) {
        const char* base = rec + (g * steps + t) * 2560;
        cc[d] = *(const u2*)(base + 8 * lane);
        va[d] = *(const d2*)(base + 512 + 16 * lane);
        vb[d] = *(const d2*)(base + 512 + 1024 + 16 * lane);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, d);
    double acc = 0; unsigned int ca = 0;
    for (size_t t = 0; t + D <= steps; t += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            acc += va[d][0] + va[d][1] + vb[d][0] + vb[d][1]; ca += cc[d][0] ^ cc[d][1];
            size_t nt = t + d + D;
            load(d, nt < steps ? nt : d);
        }
    }
    if (acc == 12345.678 && ca == 77) *sink = 1;
}

template <typename F> float time_ms(F f, int reps) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    size_t mb = argc > 1 ? atol(argv[1]) : 512;
    size_t bytes = mb << 20, n = bytes / 16;
    f4* p; float* sink;
    CHECK(hipMalloc(&p, bytes + (1 << 20))); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(p, 1, bytes + (1 << 20)));
    auto rep = [&](const char* name, float ms, size_t b) { printf("%-28s %8.1f us  %8.1f GB/s\n", name, ms * 1e3, b / ms / 1e6); };
    rep("seq U=4 grid=2048", time_ms([&] { k_seq<4><<<2048, 256>>>(p, n, sink); }, 20), bytes);
    rep("seq U=8 grid=2048", time_ms([&] { k_seq<8><<<2048, 256>>>(p, n, sink); }, 20), bytes);
    rep("seq U=8 grid=4096", time_ms([&] { k_seq<8><<<4096, 256>>>(p, n, sink); }, 20), bytes);
    rep("wgchunk 24KiB/WG", time_ms([&] { k_wgchunk<<<(unsigned)((n + 1535) / 1536), 256>>>(p, n, sink); }, 20), bytes);
    const int blocks = 512;
    size_t len = n / (blocks * 8) / 64 * 64;   // f4 elements per wave, multiple of 64
    size_t b2 = len * blocks * 8 * 16;
#define WS(D) rep("wavestream D=" #D, time_ms([&] { k_wavestream<D, false><<<blocks, 512>>>(p, len, sink); }, 20), b2); \
              rep("wginter    D=" #D, time_ms([&] { k_wavestream<D, true><<<blocks, 512>>>(p, len, sink); }, 20), b2);
    WS(2) WS(4) WS(8) WS(12) WS(16)
    {   // alternate direction between passes (Infinity Cache reuse?)
        int pass = 0;
        rep("wavestream D=4 same direction", time_ms([&] { k_wavestream_dir<4><<<blocks, 512>>>(p, len, 0, sink); }, 20), b2);
        rep("wavestream D=4 alternating", time_ms([&] { k_wavestream_dir<4><<<blocks, 512>>>(p, len, (pass++) & 1, sink); }, 20), b2);
        rep("wavestream D=12 same direction", time_ms([&] { k_wavestream_dir<12><<<blocks, 512>>>(p, len, 0, sink); }, 20), b2);
        rep("wavestream D=12 alternating", time_ms([&] { k_wavestream_dir<12><<<blocks, 512>>>(p, len, (pass++) & 1, sink); }, 20), b2);
    }
    {
        size_t steps = (size_t)36121058 / (blocks * 8) / 256;   // class-C sized: 36.1 M entries
        size_t entries = steps * blocks * 8 * 256;
        double* val; unsigned short* col;
        CHECK(hipMalloc(&val, entries * 8 + 65536)); CHECK(hipMalloc(&col, entries * 2 + 65536));
        CHECK(hipMemset(val, 0, entries * 8)); CHECK(hipMemset(col, 0, entries * 2));
        size_t b3 = entries * 10;
#define TS(D, SH) rep("tileshape D=" #D " shape=" #SH, time_ms([&] { k_tileshape<D, SH><<<blocks, 512>>>(val, col, steps, sink); }, 20), b3);
        TS(2, 0) TS(4, 0) TS(6, 0) TS(2, 1) TS(4, 1) TS(6, 1) TS(2, 2) TS(4, 2) TS(6, 2)
        char* rec; CHECK(hipMalloc(&rec, entries * 10 + 65536)); CHECK(hipMemset(rec, 0, entries * 10));
#define RS(D) rep("recshape  D=" #D " (one interleaved stream)", time_ms([&] { k_recshape<D><<<blocks, 512>>>(rec, steps, sink); }, 20), b3);
        RS(2) RS(4) RS(6)
    }
    return 0;
}
