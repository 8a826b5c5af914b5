// tools/xferbench.hip -- how should a vector of a few hundred KB .. tens of MB cross PCIe on every
// call of the ABI: the DMA path (hipMemcpyAsync on pinned memory) or a kernel that reads / writes
// the pinned host buffer directly (zero-copy)?  One stream, a sync per round trip, like the harness.
//   hipcc --offload-arch=gfx950 -O3 tools/xferbench.hip -o /tmp/xb && /tmp/xb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename F> static double time_us(F f, int reps)
{
    f();
    CHECK(hipDeviceSynchronize());
    double t0 = now();
    for (int i = 0; i < reps; ++i)
        f();
    return (now() - t0) / reps * 1e6;
}

int main()
{
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t maxb = (size_t)48 << 20;
    void *hx, *hy, *dx, *dy;
    CHECK(hipHostMalloc(&hx, maxb, hipHostMallocDefault));
    CHECK(hipHostMalloc(&hy, maxb, hipHostMallocDefault));
    void* rx = aligned_alloc(4096, maxb); // registered malloc memory, like a Fortran COMMON block pinned in place
    void* ry = aligned_alloc(4096, maxb);
    memset(rx, 1, maxb), memset(ry, 1, maxb);
    CHECK(hipHostRegister(rx, maxb, hipHostRegisterDefault));
    CHECK(hipHostRegister(ry, maxb, hipHostRegisterDefault));
    void *rxd, *ryd;
    CHECK(hipHostGetDevicePointer(&rxd, rx, 0));
    CHECK(hipHostGetDevicePointer(&ryd, ry, 0));
    CHECK(hipMalloc(&dx, maxb));
    CHECK(hipMalloc(&dy, maxb));
    memset(hx, 1, maxb);
    printf("%10s | %9s %9s %9s %9s | %9s %9s %9s %9s | %9s %9s\n", "bytes", "h2d dma", "h2d kern", "h2d dmaR", "h2d kernR",
           "d2h dma", "d2h kern", "d2h dmaR", "d2h kernR", "rt dma", "rt kern");
    const size_t sizes[] = {(size_t)112 << 10, (size_t)600 << 10, (size_t)1200 << 10, (size_t)4 << 20, (size_t)12 << 20, (size_t)39 << 20};
    for (size_t b : sizes) {
        const size_t n16 = b / 16;
        const int grid = (int)((n16 + 255) / 256 < 1024 ? (n16 + 255) / 256 : 1024);
        const int reps = b > ((size_t)8 << 20) ? 50 : 300;
        double a1 = time_us([&] { CHECK(hipMemcpyAsync(dx, hx, b, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st)); }, reps);
        double a2 = time_us([&] { k_copy<<<grid, 256, 0, st>>>((const f4*)hx, (f4*)dx, n16); CHECK(hipStreamSynchronize(st)); }, reps);
        double a3 = time_us([&] { CHECK(hipMemcpyAsync(dx, rx, b, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st)); }, reps);
        double a4 = time_us([&] { k_copy<<<grid, 256, 0, st>>>((const f4*)rxd, (f4*)dx, n16); CHECK(hipStreamSynchronize(st)); }, reps);
        double b1 = time_us([&] { CHECK(hipMemcpyAsync(hy, dy, b, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st)); }, reps);
        double b2 = time_us([&] { k_copy<<<grid, 256, 0, st>>>((const f4*)dy, (f4*)hy, n16); CHECK(hipStreamSynchronize(st)); }, reps);
        double b3 = time_us([&] { CHECK(hipMemcpyAsync(ry, dy, b, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st)); }, reps);
        double b4 = time_us([&] { k_copy<<<grid, 256, 0, st>>>((const f4*)dy, (f4*)ryd, n16); CHECK(hipStreamSynchronize(st)); }, reps);
        // round trip as in the harness: x in, (a kernel in between: device copy dx->dy), y out, one sync
        double c1 = time_us([&] {
            CHECK(hipMemcpyAsync(dx, rx, b, hipMemcpyHostToDevice, st));
            k_copy<<<grid, 256, 0, st>>>((const f4*)dx, (f4*)dy, n16);
            CHECK(hipMemcpyAsync(ry, dy, b, hipMemcpyDeviceToHost, st));
            CHECK(hipStreamSynchronize(st)); }, reps);
        double c2 = time_us([&] {
            k_copy<<<grid, 256, 0, st>>>((const f4*)rxd, (f4*)dx, n16);
            k_copy<<<grid, 256, 0, st>>>((const f4*)dx, (f4*)ryd, n16);
            CHECK(hipStreamSynchronize(st)); }, reps);
        printf("%10zu | %9.1f %9.1f %9.1f %9.1f | %9.1f %9.1f %9.1f %9.1f | %9.1f %9.1f   (us)\n", b, a1, a2, a3, a4, b1, b2, b3, b4, c1, c2);
    }
    return 0;
}
