// tools/pbbench.hip -- feasibility of a column-BLOCKED product for matrices whose x does not fit a cache that gathers
// well (SparseBench 170^3: x = 39 MB, columns uniform; STREAM runs it at the Infinity Cache's gather rate, 431 us).
//   phase 1: the matrix re-ordered block-major -- column block c = columns [c*CB, (c+1)*CB), its entries in CSR order --
//            so that everything an XCD gathers at one time comes from one x block that fits its L2; every (row, block)
//            segment is summed and written to a compact array, in order;
//   phase 2: y[r] = sum over blocks of the row's segment sums: per 64 rows one contiguous read per block.
// Synthetic matrix of SparseBench's shape, generated here; both phases timed; y checked against a host CSR product.
//   hipcc --offload-arch=gfx950 -O3 tools/pbbench.hip -o /tmp/pbbench && /tmp/pbbench [size] [nblk] [chunk] [wgs_per_cu] [U]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// one wave per chunk; chunk k = entries [cs[k], cs[k+1]) of the block-major arrays, cut at segment starts.
// U groups of 64 entries are loaded (and their gathers issued) before the first of them is scanned.
template <int U, bool SCAN>
__global__ void __launch_bounds__(256) k_phase1(const double* __restrict__ val, const int* __restrict__ col,
                                                const uint64_t* __restrict__ starts, const int64_t* __restrict__ cs,
                                                const int64_t* __restrict__ sb, const int* __restrict__ blk_chunk0,
                                                int nblk, const double* __restrict__ x, double* __restrict__ seg)
{
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;
    const int wave_in_xcd = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int waves_per_xcd = (gridDim.x >> 3) * 4;
    for (int c = xcd; c < nblk; c += 8) {
        const int k0 = blk_chunk0[c], k1 = blk_chunk0[c + 1];
        for (int k = k0 + wave_in_xcd; k < k1; k += waves_per_xcd) {
            const int64_t e0 = cs[k], e1 = cs[k + 1];
            int64_t out = sb[k];
            double carry = 0.0;
            for (int64_t eb = e0; eb < e1; eb += 64 * U) {
                double a[U];
                int ci[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t me = eb + u * 64 + lane;
                    const bool ok = me < e1;
                    a[u] = ok ? val[me] : 0.0;
                    ci[u] = ok ? col[me] : 0;
                }
                double v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[u] = a[u] * x[ci[u]];
                if constexpr (!SCAN) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (eb + u * 64 + lane < e1)
                            seg[eb + u * 64 + lane] = v[u];
                    continue;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t e = eb + u * 64;
                    if (e >= e1)
                        break;
                    const int nvalid = (int)((e1 - e) < 64 ? (e1 - e) : 64);
                    // 65-bit window of start flags at bit offset e: bit i = entry e+i starts a segment
                    const int64_t w = e >> 6;
                    const int sh = (int)(e & 63);
                    uint64_t lo = starts[w], hi = starts[w + 1], hi2 = starts[w + 2];
                    uint64_t st = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                    uint64_t nx = sh ? (hi >> sh) | (hi2 << (64 - sh)) : hi; // bit 0 = entry e+64
                    uint64_t valid = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
                    st &= valid;
                    uint64_t en = ((st >> 1) | (nx << 63)) & valid;
                    if (nvalid < 64)
                        en |= 1ull << (nvalid - 1); // the chunk ends at a segment start
                    uint64_t below = st & ((lanes_below() << 1) | 1ull);
                    int s0 = below ? 63 - __builtin_clzll(below) : 0;
                    bool continues = (below == 0);
                    double t = v[u];
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        double up = __shfl_up(t, d, 64);
                        if (lane - d >= s0)
                            t += up;
                    }
                    if (continues)
                        t += carry;
                    if ((en >> lane) & 1ull)
                        seg[out + __builtin_popcountll(en & lanes_below())] = t;
                    out += __builtin_popcountll(en);
                    double last = __shfl(t, nvalid - 1, 64);
                    carry = ((en >> (nvalid - 1)) & 1ull) ? 0.0 : last;
                }
            }
        }
    }
}

// lane = row; mask[r] bit c = the row has a segment in block c; base2[(r/64)*nblk + c] = index of the first of them.
// All bases of the 64 rows come with one load, all segment sums are requested before the first is added.
template <int NBLK>
__global__ void __launch_bounds__(256) k_phase2(const uint32_t* __restrict__ mask, const int64_t* __restrict__ base2,
                                                int nblk, const double* __restrict__ seg, double* __restrict__ y, int rows)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int chunk = r >> 6;
    const uint32_t m = r < rows ? mask[r] : 0u;
    const int64_t myb = lane < NBLK ? base2[(int64_t)chunk * NBLK + lane] : 0;
    double sv[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
        const bool has = (m >> c) & 1u;
        const uint64_t b = __ballot(has);
        const int64_t base = __shfl(myb, c, 64);
        sv[c] = has ? seg[base + __builtin_popcountll(b & lanes_below())] : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
        if ((m >> c) & 1u)
            acc += sv[c];
    if (r < rows)
        y[r] = acc;
}

static int g_scan = 1;
template <int U> static void launch1(int grid, const double* val, const int* col, const uint64_t* starts, const int64_t* cs,
                                    const int64_t* sb, const int* bc0, int nblk, const double* x, double* seg)
{
    if (g_scan)
        k_phase1<U, true><<<grid, 256>>>(val, col, starts, cs, sb, bc0, nblk, x, seg);
    else
        k_phase1<U, false><<<grid, 256>>>(val, col, starts, cs, sb, bc0, nblk, x, seg);
}
static int g_u = 4;
#define PHASE1(grid, ...) do { if (g_u == 1) launch1<1>(grid, __VA_ARGS__); else if (g_u == 2) launch1<2>(grid, __VA_ARGS__); \
    else if (g_u == 4) launch1<4>(grid, __VA_ARGS__); else launch1<8>(grid, __VA_ARGS__); } while (0)
#define PHASE2(grid, ...) do { if (nblk == 8) k_phase2<8><<<grid, 256>>>(__VA_ARGS__); else if (nblk == 16) k_phase2<16><<<grid, 256>>>(__VA_ARGS__); \
    else if (nblk == 24) k_phase2<24><<<grid, 256>>>(__VA_ARGS__); else k_phase2<32><<<grid, 256>>>(__VA_ARGS__); } while (0)


// probes: MODE 0 = gathers of the block-major column stream only (no values, no stores), XCD-private blocks;
//         MODE 1 = the same, every XCD on every block (blocks walked in the same order by everybody);
//         MODE 2 = MODE 0 + the value stream; MODE 3 = MODE 2 + one 8-byte store per entry
template <int MODE>
__global__ void __launch_bounds__(256) k_probe(const double* __restrict__ val, const int* __restrict__ col,
                                               const int64_t* __restrict__ bcount, int nblk, const double* __restrict__ x,
                                               double* __restrict__ seg, double* __restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;
    double acc = 0.0;
    const int cstep = MODE == 1 ? 1 : 8;
    const int64_t wave = MODE == 1 ? (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6) : (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = MODE == 1 ? (int64_t)gridDim.x * 4 : (gridDim.x >> 3) * 4;
    for (int c = MODE == 1 ? 0 : xcd; c < nblk; c += cstep) {
        const int64_t b0 = bcount[c], b1 = bcount[c + 1];
        for (int64_t e = b0 + wave * 256; e < b1; e += nwaves * 256) {
            int ci[4];
            double a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t me = e + u * 64 + lane;
                ci[u] = me < b1 ? col[me] : 0;
                a[u] = (MODE >= 2 && me < b1) ? val[me] : 1.0;
            }
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = a[u] * x[ci[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 3) {
                    int64_t me = e + u * 64 + lane;
                    if (me < b1)
                        seg[me] = v[u];
                } else {
                    acc += v[u];
                }
            }
        }
    }
    if (MODE != 3)
        sink[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// ---- design B: no segment arrays, no second kernel -------------------------------------------------------------------
// A wave owns RW = 2^RB consecutive rows, its accumulators sit in LDS; its entries are stored contiguously, ordered by
// column block (CSR order inside a block), each with one 32-bit word (local row << (32-RB) ... | absolute column).  All
// waves walk their streams front to back, so at any time the whole chip gathers from about one column block.
template <int U, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_blockedB(const double* __restrict__ val, const uint32_t* __restrict__ pk,
                                                        const int64_t* __restrict__ rstart, int rb, int nranges,
                                                        const double* __restrict__ x, double* __restrict__ y, int rows)
{
    extern __shared__ double yacc[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int range = blockIdx.x * WAVES + w;
    const int rw = 1 << rb;
    double* acc = yacc + (size_t)w * rw;
    for (int i = lane; i < rw; i += 64)
        acc[i] = 0.0;
    if (range >= nranges)
        return;
    const int64_t s0 = rstart[range], s1 = rstart[range + 1];
    const uint32_t cmask = (1u << (32 - rb)) - 1u;
    for (int64_t eb = s0; eb < s1; eb += 64 * U) {
        double a[U];
        uint32_t q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t me = eb + u * 64 + lane;
            const bool ok = me < s1;
            a[u] = ok ? val[me] : 0.0;
            q[u] = ok ? pk[me] : 0u;
        }
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = a[u] * x[q[u] & cmask];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (eb + u * 64 + lane < s1)
                __hip_atomic_fetch_add(&acc[q[u] >> (32 - rb)], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    const int r0 = range << rb;
    for (int i = lane; i < rw && r0 + i < rows; i += 64)
        y[r0 + i] = acc[i];
}

// probe 4/5: like MODE 3 (values + gathers + stores, XCD-private blocks) with non-temporal streams (4) and with the
// next iteration's streams requested before this iteration's gathers are waited for (5, also non-temporal)
template <int MODE>
__global__ void __launch_bounds__(256) k_probe_nt(const double* __restrict__ val, const int* __restrict__ col,
                                                  const int64_t* __restrict__ bcount, int nblk, const double* __restrict__ x,
                                                  double* __restrict__ seg)
{
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;
    const int64_t wave = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (gridDim.x >> 3) * 4;
    for (int c = xcd; c < nblk; c += 8) {
        const int64_t b0 = bcount[c], b1 = bcount[c + 1];
        int ci[4], cn[4];
        double a[4], an[4];
        int64_t e = b0 + wave * 256;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t me = e + u * 64 + lane;
            ci[u] = me < b1 ? __builtin_nontemporal_load(col + me) : 0;
            a[u] = me < b1 ? __builtin_nontemporal_load(val + me) : 0.0;
        }
        for (; e < b1; e += nwaves * 256) {
            if (MODE == 5) {
                const int64_t en = e + nwaves * 256;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int64_t me = en + u * 64 + lane;
                    cn[u] = me < b1 ? __builtin_nontemporal_load(col + me) : 0;
                    an[u] = me < b1 ? __builtin_nontemporal_load(val + me) : 0.0;
                }
            }
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = a[u] * x[ci[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t me = e + u * 64 + lane;
                if (me < b1)
                    __builtin_nontemporal_store(v[u], seg + me);
            }
            if (MODE == 5) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ci[u] = cn[u];
                    a[u] = an[u];
                }
            } else {
                const int64_t en = e + nwaves * 256;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int64_t me = en + u * 64 + lane;
                    ci[u] = me < b1 ? __builtin_nontemporal_load(col + me) : 0;
                    a[u] = me < b1 ? __builtin_nontemporal_load(val + me) : 0.0;
                }
            }
        }
    }
}

int main(int argc, char** argv)
{
    const int size = argc > 1 ? atoi(argv[1]) : 170;
    const int nblk = argc > 2 ? atoi(argv[2]) : 16;
    const int chunk = argc > 3 ? atoi(argv[3]) : 512;
    const int wgs_per_cu = argc > 4 ? atoi(argv[4]) : 8;
    g_u = argc > 5 ? atoi(argv[5]) : 4;
    g_scan = argc > 6 ? atoi(argv[6]) : 1; // 0: products written as they are (no segment sums; y is then wrong)
    const int rows = size * size * size, cols = rows;
    std::mt19937_64 rng(170);
    std::normal_distribution<double> nlen(5.0, 4.0), nval(0.0, 2.0);
    std::vector<int64_t> rowptr(rows + 1, 0);
    std::vector<int> ci;
    std::vector<double> a;
    ci.reserve((size_t)rows * 6);
    for (int r = 0; r < rows; ++r) {
        int len = std::max(1, (int)nlen(rng));
        size_t b = ci.size();
        for (int i = 0; i < len; ++i)
            ci.push_back((int)(rng() % (uint64_t)cols));
        std::sort(ci.begin() + b, ci.end());
        rowptr[r + 1] = (int64_t)ci.size();
    }
    const int64_t nnz = (int64_t)ci.size();
    a.resize(nnz);
    for (auto& v : a)
        v = std::fabs(nval(rng));
    std::vector<double> x(cols);
    for (auto& v : x)
        v = nval(rng);
    printf("rows %d nnz %lld  nblk %d  chunk %d\n", rows, (long long)nnz, nblk, chunk);

    // ---- block-major re-order ------------------------------------------------------------------
    const int cb = ((cols + nblk - 1) / nblk + 63) & ~63;
    std::vector<int64_t> bcount(nblk + 1, 0);
    for (int64_t e = 0; e < nnz; ++e)
        bcount[ci[e] / cb + 1]++;
    for (int c = 0; c < nblk; ++c)
        bcount[c + 1] += bcount[c];
    std::vector<double> bval(nnz + 64);
    std::vector<int> bcol(nnz + 64, 0), brow(nnz);
    {
        std::vector<int64_t> pos(bcount.begin(), bcount.end() - 1);
        for (int r = 0; r < rows; ++r)
            for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
                int64_t p = pos[ci[e] / cb]++;
                bval[p] = a[e];
                bcol[p] = ci[e];
                brow[p] = r;
            }
    }
    std::vector<uint64_t> starts(nnz / 64 + 4, 0);
    std::vector<uint32_t> mask(rows, 0);
    std::vector<int64_t> cs, sb;
    std::vector<int> blk_chunk0(nblk + 1, 0);
    std::vector<int64_t> base2((size_t)((rows + 63) / 64) * nblk, 0);
    int64_t nseg = 0;
    for (int c = 0; c < nblk; ++c) {
        blk_chunk0[c] = (int)cs.size();
        int64_t next_cut = bcount[c];
        int last_chunk64 = -1;
        for (int64_t e = bcount[c]; e < bcount[c + 1]; ++e) {
            bool st = (e == bcount[c]) || brow[e] != brow[e - 1];
            if (st) {
                starts[e >> 6] |= 1ull << (e & 63);
                if (e >= next_cut) {
                    cs.push_back(e);
                    sb.push_back(nseg);
                    next_cut = e + chunk;
                }
                mask[brow[e]] |= 1u << c;
                int ch = brow[e] >> 6;
                if (ch != last_chunk64) {
                    base2[(size_t)ch * nblk + c] = nseg;
                    last_chunk64 = ch;
                }
                ++nseg;
            }
        }
    }
    blk_chunk0[nblk] = (int)cs.size();
    cs.push_back(nnz);
    sb.push_back(nseg);
    starts[nnz >> 6] |= 1ull << (nnz & 63); // sentinel: the entry after the last one "starts"
    printf("segments %lld (%.2f entries each), chunks %zu\n", (long long)nseg, (double)nnz / nseg, cs.size() - 1);

    double *d_val, *d_x, *d_seg, *d_y;
    int *d_col, *d_bc0;
    uint64_t* d_starts;
    int64_t *d_cs, *d_sb, *d_base2;
    uint32_t* d_mask;
    CHECK(hipMalloc(&d_val, bval.size() * 8));
    CHECK(hipMalloc(&d_col, bcol.size() * 4));
    CHECK(hipMalloc(&d_starts, starts.size() * 8));
    CHECK(hipMalloc(&d_cs, cs.size() * 8));
    CHECK(hipMalloc(&d_sb, sb.size() * 8));
    CHECK(hipMalloc(&d_bc0, blk_chunk0.size() * 4));
    CHECK(hipMalloc(&d_x, (size_t)cols * 8));
    CHECK(hipMalloc(&d_seg, (size_t)(nnz + 64) * 8));
    CHECK(hipMalloc(&d_y, (size_t)rows * 8));
    CHECK(hipMalloc(&d_mask, (size_t)rows * 4));
    CHECK(hipMalloc(&d_base2, base2.size() * 8));
    CHECK(hipMemcpy(d_val, bval.data(), bval.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_col, bcol.data(), bcol.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_starts, starts.data(), starts.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_cs, cs.data(), cs.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_sb, sb.data(), sb.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_bc0, blk_chunk0.data(), blk_chunk0.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_x, x.data(), (size_t)cols * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_mask, mask.data(), (size_t)rows * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_base2, base2.data(), base2.size() * 8, hipMemcpyHostToDevice));

    const int grid1 = 256 * wgs_per_cu, grid2 = (rows + 255) / 256;
    hipEvent_t ev[4];
    for (auto& evt : ev)
        CHECK(hipEventCreate(&evt));
    for (int i = 0; i < 5; ++i) {
        PHASE1(grid1, d_val, d_col, d_starts, d_cs, d_sb, d_bc0, nblk, d_x, d_seg);
        PHASE2(grid2, d_mask, d_base2, nblk, d_seg, d_y, rows);
    }
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    {
        int64_t* d_bcount;
        double* d_sink;
        CHECK(hipMalloc(&d_bcount, bcount.size() * 8));
        CHECK(hipMalloc(&d_sink, (size_t)grid1 * 256 * 8));
        CHECK(hipMemcpy(d_bcount, bcount.data(), bcount.size() * 8, hipMemcpyHostToDevice));
        const char* names[6] = {"gathers only, XCD-private blocks", "gathers only, all XCDs on each block", "+ value stream", "+ 8-byte store per entry",
                                "all three, streams non-temporal", "all three, non-temporal + prefetched"};
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e30f;
            for (int i = 0; i < 8; ++i) {
                CHECK(hipEventRecord(ev[0]));
                if (mode == 0) k_probe<0><<<grid1, 256>>>(d_val, d_col, d_bcount, nblk, d_x, d_seg, d_sink);
                if (mode == 1) k_probe<1><<<grid1, 256>>>(d_val, d_col, d_bcount, nblk, d_x, d_seg, d_sink);
                if (mode == 2) k_probe<2><<<grid1, 256>>>(d_val, d_col, d_bcount, nblk, d_x, d_seg, d_sink);
                if (mode == 3) k_probe<3><<<grid1, 256>>>(d_val, d_col, d_bcount, nblk, d_x, d_seg, d_sink);
                if (mode == 4) k_probe_nt<4><<<grid1, 256>>>(d_val, d_col, d_bcount, nblk, d_x, d_seg);
                if (mode == 5) k_probe_nt<5><<<grid1, 256>>>(d_val, d_col, d_bcount, nblk, d_x, d_seg);
                CHECK(hipEventRecord(ev[1]));
                CHECK(hipEventSynchronize(ev[1]));
                float m;
                CHECK(hipEventElapsedTime(&m, ev[0], ev[1]));
                best = m < best ? m : best;
            }
            printf("probe %-40s %.1f us  %.1f G gathers/s\n", names[mode], best * 1e3, nnz / (best * 1e-3) / 1e9);
        }
    }
    float t1 = 0, t2 = 0, tt = 0;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(ev[0]));
        PHASE1(grid1, d_val, d_col, d_starts, d_cs, d_sb, d_bc0, nblk, d_x, d_seg);
        CHECK(hipEventRecord(ev[1]));
        PHASE2(grid2, d_mask, d_base2, nblk, d_seg, d_y, rows);
        CHECK(hipEventRecord(ev[2]));
        CHECK(hipEventSynchronize(ev[2]));
        float m;
        CHECK(hipEventElapsedTime(&m, ev[0], ev[1]));
        t1 += m;
        CHECK(hipEventElapsedTime(&m, ev[1], ev[2]));
        t2 += m;
    }
    CHECK(hipEventRecord(ev[0]));
    for (int i = 0; i < reps; ++i) {
        PHASE1(grid1, d_val, d_col, d_starts, d_cs, d_sb, d_bc0, nblk, d_x, d_seg);
        PHASE2(grid2, d_mask, d_base2, nblk, d_seg, d_y, rows);
    }
    CHECK(hipEventRecord(ev[1]));
    CHECK(hipEventSynchronize(ev[1]));
    CHECK(hipEventElapsedTime(&tt, ev[0], ev[1]));
    std::vector<double> y(rows);
    CHECK(hipMemcpy(y.data(), d_y, (size_t)rows * 8, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (int r = 0; r < rows; ++r) {
        double s = 0, ab = 0;
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            s += a[e] * x[ci[e]];
            ab += std::fabs(a[e] * x[ci[e]]);
        }
        if (!(std::fabs(s - y[r]) <= 64 * 2.3e-16 * ab))
            if (bad++ < 5)
                printf("row %d: got %.17g want %.17g\n", r, y[r], s);
    }
    const double bytes = (double)nnz * 12 + (rows + 1) * 4.0 + cols * 8.0 + rows * 8.0;
    printf("phase1 %.1f us  phase2 %.1f us  back-to-back %.1f us  (%.0f GB/s algorithmic, %.1f G gathers/s in phase 1)  bad rows %lld\n",
           t1 / reps * 1e3, t2 / reps * 1e3, tt / reps * 1e3, bytes / (tt / reps * 1e-3) / 1e9,
           nnz / (t1 / reps * 1e-3) / 1e9, (long long)bad);

    // ---- design B ----
    {
        int colbits = 1;
        while ((1ll << colbits) < cols)
            ++colbits;
        const int rb = std::min(10, 32 - colbits);
        const int rw = 1 << rb, nranges = (rows + rw - 1) / rw;
        std::vector<int64_t> rstart(nranges + 1, 0);
        std::vector<double> v2(nnz + 64);
        std::vector<uint32_t> pk(nnz + 64, 0);
        int64_t p = 0;
        std::vector<int64_t> cnt(nblk + 1);
        for (int g = 0; g < nranges; ++g) {
            rstart[g] = p;
            const int r0 = g * rw, r1 = std::min(rows, r0 + rw);
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int64_t e = rowptr[r0]; e < rowptr[r1]; ++e)
                cnt[ci[e] / cb + 1]++;
            for (int c = 0; c < nblk; ++c)
                cnt[c + 1] += cnt[c];
            for (int r = r0; r < r1; ++r)
                for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
                    int64_t d = p + cnt[ci[e] / cb]++;
                    v2[d] = a[e];
                    pk[d] = ((uint32_t)(r - r0) << (32 - rb)) | (uint32_t)ci[e];
                }
            p += rowptr[r1] - rowptr[r0];
        }
        rstart[nranges] = p;
        double* d_v2;
        uint32_t* d_pk;
        int64_t* d_rs;
        CHECK(hipMalloc(&d_v2, v2.size() * 8));
        CHECK(hipMalloc(&d_pk, pk.size() * 4));
        CHECK(hipMalloc(&d_rs, rstart.size() * 8));
        CHECK(hipMemcpy(d_v2, v2.data(), v2.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_rs, rstart.data(), rstart.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipMemset(d_y, 0, (size_t)rows * 8));
        for (int variant = 0; variant < 4; ++variant) {
            const int waves = (variant & 1) ? 8 : 4, u = (variant & 2) ? 8 : 4;
            const int grid = (nranges + waves - 1) / waves;
            const size_t lds = (size_t)waves * rw * 8;
            float best = 1e30f;
            for (int i = 0; i < 10; ++i) {
                CHECK(hipEventRecord(ev[0]));
                if (waves == 4 && u == 4) k_blockedB<4, 4><<<grid, 256, lds>>>(d_v2, d_pk, d_rs, rb, nranges, d_x, d_y, rows);
                if (waves == 8 && u == 4) k_blockedB<4, 8><<<grid, 512, lds>>>(d_v2, d_pk, d_rs, rb, nranges, d_x, d_y, rows);
                if (waves == 4 && u == 8) k_blockedB<8, 4><<<grid, 256, lds>>>(d_v2, d_pk, d_rs, rb, nranges, d_x, d_y, rows);
                if (waves == 8 && u == 8) k_blockedB<8, 8><<<grid, 512, lds>>>(d_v2, d_pk, d_rs, rb, nranges, d_x, d_y, rows);
                CHECK(hipEventRecord(ev[1]));
                CHECK(hipEventSynchronize(ev[1]));
                float m;
                CHECK(hipEventElapsedTime(&m, ev[0], ev[1]));
                best = m < best ? m : best;
            }
            CHECK(hipMemcpy(y.data(), d_y, (size_t)rows * 8, hipMemcpyDeviceToHost));
            int64_t badB = 0;
            for (int r = 0; r < rows; ++r) {
                double sum = 0, ab = 0;
                for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
                    sum += a[e] * x[ci[e]];
                    ab += std::fabs(a[e] * x[ci[e]]);
                }
                if (!(std::fabs(sum - y[r]) <= 64 * 2.3e-16 * ab))
                    ++badB;
            }
            printf("design B  rows/wave %d  waves/WG %d  U %d  grid %d: %.1f us  (%.0f GB/s algorithmic, %.1f G gathers/s)  bad rows %lld\n",
                   rw, waves, u, grid, best * 1e3, bytes / (best * 1e-3) / 1e9, nnz / (best * 1e-3) / 1e9, (long long)badB);
        }
    }
    return bad != 0;
}
