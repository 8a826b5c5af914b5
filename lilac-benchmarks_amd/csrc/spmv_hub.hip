/*
 * csrc/spmv_hub.hip -- the HUB plan: CSR y = A x for power-law matrices (the adjacency matrices bfs/library.cc:268-279 and
 * pagerank/main.cpp:128-149 multiply once per level / iteration), where a few thousand "hub" columns of x take most of the
 * gathers.  The arithmetic is libspmv/native-impl.c:1-25's; the order of the additions is this file's (fixed by the layout,
 * no atomics on values).
 *
 * Why (profiles/r04_stream_ablation_rmat.log, r04_stream_sq_counters.json): on R-MAT 2^20 x 16 the row-segment kernel
 * (k_stream + the long-row kernels, 107 us) spends 3 us on its per-row work and 54 us on gather MISSES -- waves stand 66 % of
 * their cycles waiting to issue the next memory instruction --, and the long rows' two kernels run 22 us behind it on the
 * same stream.  16 384 of the graph's 1 M columns hold 58 % of its entries.
 *
 * The plan:
 *   - the matrix stays where it is: values are read from the CSR copy (a value refresh costs this plan nothing); a second
 *     copy of the COLUMN INDICES (col2, 4 bytes per entry, read instead of the original) says per entry: hub -> the slot of an
 *     LDS table (bit 31), otherwise the 0-based column; bit 30 marks the last entry of a row;
 *   - a second tier behind the LDS table (round 4, profiles/r04_hub_table_by_size.log): the next 256 K most referenced columns
 *     get slots of a dense 2-MB table in memory (col2 bit 29).  A gather that misses the cache moves a 128-byte line for
 *     8 bytes; the hot columns of a power-law graph are scattered one to a line over all of x, packed they are lines that every
 *     XCD's L2 keeps.  R-MAT 2^22 (x = 32 MB): 520 -> 390 us; 2^20 (x = 8 MB, half of it per L2 anyway): 72 -> 70;
 *   - k_hub_pack gathers the hub entries of x into the dense tables (one small launch);
 *   - k_hub: one persistent workgroup of 16 waves per CU copies the table into LDS (128 KB) and takes a contiguous range of
 *     CHUNKS of the entry stream -- 1024 entries each, whatever the rows: no row blocks, no long-row kernels, every wave
 *     the same amount of work; its waves draw chunks from an LDS ticket.  Per sub-step of 64 x 16 bytes a wave multiplies
 *     its entries by x from LDS (hubs) or from memory (the rest: the only gathers left), sums rows with the segmented
 *     DPP scan of spmv_wave.h, parks finished row sums in its LDS ring and writes them out 64 rows at a time; row ends are
 *     numbered from the chunk's rank0, rows without entries are skipped through a list of the others;
 *   - a row that crosses a chunk boundary leaves one carry per chunk; k_hub_fix adds them to the piece in the row's last
 *     chunk, in chunk order (a wave per such row: a row of 10^8 entries has 10^5 carries), and zeroes the empty rows.
 * Layout decisions: spmv_hub_plan.c (CPU-testable).  Built on the device from the CSR copy.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "spmv_internal.h"
#include "spmv_wave.h"

namespace {

constexpr unsigned int HUB = SPMV_HUB_FLAG_HUB, END = SPMV_HUB_FLAG_END, TBL = SPMV_HUB_FLAG_TBL, IDX = 0x1fffffffu;
constexpr int WAVES = SPMV_HUB_WAVES, NT_WG = WAVES * 64, CH = SPMV_HUB_CHUNK;

template <typename T> struct HubGeom {
    static constexpr int E = Vec16<T>::E;          /* entries per lane per sub-step: one 16-byte value load         */
    static constexpr int SUB = 64 * E;             /* entries per sub-step: 128 / 256                               */
    static constexpr int NSUB = CH / SUB;          /* sub-steps per chunk: 8 / 4                                    */
    static constexpr int H = 4;                    /* sub-steps per pipeline stage (their loads travel together)    */
    static constexpr int NST = NSUB / H;           /* stages per chunk: 2 / 1                                       */
    static constexpr int RING = 2 * SUB;           /* row sums a wave parks before they leave: 2 KB per wave        */
    static constexpr int SLOTS = sizeof(T) == 8 ? 16352 : 32736; /* = spmv_hub_lds_slots: table + rings + ticket <= 160 KB */
    static_assert(NSUB % H == 0 && NST >= 1, "a chunk is a whole number of stages");
};

/* one pipeline stage of a wave in registers: H sub-steps, E consecutive entries per lane each */
template <typename T> struct HubStage {
    typename Vec16<T>::idx_t c[HubGeom<T>::H];
    typename Vec16<T>::val_t v[HubGeom<T>::H];
};

template <typename T>
__device__ __forceinline__ void hub_load(HubStage<T> &s, const unsigned int *__restrict__ col2, const T *__restrict__ val, int chunk,
                                         int st, int lane)
{
    using G = HubGeom<T>;
    using VT = typename Vec16<T>::val_t;
    using IT = typename Vec16<T>::idx_t;
    const unsigned int base = (unsigned int)chunk * (unsigned int)CH + (unsigned int)(st * G::H * G::SUB) + (unsigned int)(lane * G::E);
#pragma unroll
    for (int k = 0; k < G::H; ++k) {
        s.c[k] = *reinterpret_cast<const IT *>(col2 + base + k * G::SUB);
        s.v[k] = *reinterpret_cast<const VT *>(val + base + k * G::SUB);
    }
}

/* Every product is rounded before it is added (no contraction): the result is a function of the layout, addition by
 * addition, whichever wave happens to draw a chunk. */
#pragma clang fp contract(off)
template <typename T, bool IDENT>
__global__ void __launch_bounds__(NT_WG) k_hub(const unsigned int *__restrict__ col2, const T *__restrict__ val,
                                               const T *__restrict__ x, const T *__restrict__ xh,
                                               const int *__restrict__ rank0, const int *__restrict__ nzrow, T *__restrict__ y,
                                               T *__restrict__ carry_out, int n_chunks, int kpad, int nwg, int per_xcd)
{
    using G = HubGeom<T>;
    using VT = typename Vec16<T>::val_t;
    constexpr int E = G::E, H = G::H, RING = G::RING;
    __shared__ __attribute__((aligned(16))) T s_x[G::SLOTS + 16];
    __shared__ T s_out[WAVES][RING];
    __shared__ unsigned int s_ticket;

    const int b = (int)blockIdx.x;
    const int q = per_xcd > 0 ? (b & 7) * per_xcd + (b >> 3) : b; /* an XCD's workgroups: neighbouring chunks (rows, y) */
    if (q >= nwg)
        return; /* the whole workgroup */
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int c0 = (int)((long long)n_chunks * q / nwg), c1 = (int)((long long)n_chunks * (q + 1) / nwg);

    /* the table of hub entries of x, packed by k_hub_pack; s_x[K] = 0 is what the stream's pads multiply with */
    for (int i = tid * E; i < kpad; i += NT_WG * E)
        *reinterpret_cast<VT *>(&s_x[i]) = *reinterpret_cast<const VT *>(xh + i);
    if (tid == 0)
        s_ticket = 0u;
    __syncthreads();

    auto draw = [&]() -> int {
        unsigned int t = 0u;
        if (lane == 0)
            t = __hip_atomic_fetch_add(&s_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return c0 + (int)__builtin_amdgcn_readfirstlane((int)t);
    };

    int c = draw();
    if (c >= c1)
        return;
    int st = 0;
    HubStage<T> cur;
    hub_load<T>(cur, col2, val, c, 0, lane);

    /* per-chunk state of the wave */
    int rank_base = __builtin_amdgcn_readfirstlane(rank0[c]);
    unsigned int row = 0, flushed = 0; /* row ends met in this chunk; how many of them have left the ring (a multiple of 64) */
    T carry = (T)0;                    /* open sum of the row that spans the sub-step boundary */
    int rnext = IDENT ? 0 : nzrow[rank_base + lane];

    /* rows [flushed, flushed + 64) of the chunk leave the ring for y: lane l writes the row that ends l-th of them */
    auto flush = [&](const unsigned int upto) __attribute__((always_inline)) {
        const unsigned int r = flushed + (unsigned int)lane;
        const int rowid = IDENT ? rank_base + (int)r : rnext;
        if (r < upto)
            y[rowid] = s_out[w][r & (RING - 1)];
        flushed += 64u;
        if (!IDENT)
            rnext = nzrow[rank_base + (int)flushed + lane]; /* on its way long before the next 64 rows are complete */
    };

    for (;;) {
        /* ---- x for the stage: the entries that are not hubs gather from memory (issued first), hubs read the table ---- */
        T xv[H][E];
#pragma unroll
        for (int k = 0; k < H; ++k)
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const unsigned int cc = (unsigned int)cur.c[k][i];
                xv[k][i] = (T)0;
                if (!(cc & HUB)) /* tier 1: the dense table in memory (cache hits); else x itself */
                    xv[k][i] = ((cc & TBL) ? xh : x)[cc & IDX];
            }
        /* ---- the next stage's stream loads: no branch around them (past the end: this chunk again, never used) ---- */
        int nc = c, nst = st + 1;
        if (nst == G::NST) {
            nc = draw();
            nst = 0;
        }
        const bool more = nc < c1;
        HubStage<T> nxt;
        hub_load<T>(nxt, col2, val, more ? nc : c, more ? nst : st, lane);
#pragma unroll
        for (int k = 0; k < H; ++k)
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const unsigned int cc = (unsigned int)cur.c[k][i];
                if (cc & HUB)
                    xv[k][i] = s_x[cc & IDX];
            }

        /* ---- the stage's sub-steps ---- */
#pragma unroll
        for (int k = 0; k < H; ++k) {
            T p[E];
            int f[E];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                p[i] = cur.v[k][i] * xv[k][i];
                f[i] = (int)(((unsigned int)cur.c[k][i] >> 30) & 1u);
            }
            /* lane-local: the sum behind the lane's last row end, and whether it has one */
            T tailv = (T)0;
            int has = 0;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                tailv = tailv + p[i]; /* the flagged entry itself still belongs to the closing row */
                if (f[i])
                    tailv = (T)0;
                has |= f[i];
            }
            /* inclusive segmented scan of (tailv, has) over the wave */
            T sv = tailv;
            int sf = has;
            segscan_step<T, 0x111, 0xf>(sv, sf); /* row_shr:1 */
            segscan_step<T, 0x112, 0xf>(sv, sf); /* row_shr:2 */
            segscan_step<T, 0x114, 0xf>(sv, sf); /* row_shr:4 */
            segscan_step<T, 0x118, 0xf>(sv, sf); /* row_shr:8 */
            segscan_step<T, 0x142, 0xa>(sv, sf); /* row_bcast:15 */
            segscan_step<T, 0x143, 0xc>(sv, sf); /* row_bcast:31 */
            if (!sf)
                sv = sv + carry; /* no row end at or before this lane: the previous sub-step's open sum */
            T cin = dpp_mov<0x138, 0xf>(sv); /* wave_shr:1: what the lane below leaves open */
            if (lane == 0)
                cin = carry;
            carry = __shfl(sv, 63, 64);
            /* row numbers: the row ends in lower lanes */
            unsigned int before = 0, nflags = 0;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const unsigned long long m = __ballot(f[i] != 0);
                before += __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0));
                nflags += (unsigned int)__builtin_popcountll(m);
            }
            /* finished rows into the ring (it holds < 64 from before and takes <= SUB more: RING = 2 SUB) */
            T run = cin;
            unsigned int idx = row + before;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                run = run + p[i];
                if (f[i]) {
                    s_out[w][idx & (RING - 1)] = run;
                    ++idx;
                    run = (T)0;
                }
            }
            row += nflags;
            __builtin_amdgcn_wave_barrier();
            while (row - flushed >= 64u)
                flush(row);
            __builtin_amdgcn_wave_barrier();
        }

        if (st == G::NST - 1) { /* the chunk is done: its last < 64 rows, and what it leaves open */
            flush(row);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0)
                carry_out[c] = carry;
        }
        if (!more)
            break;
        if (nst == 0) {
            c = nc;
            rank_base = __builtin_amdgcn_readfirstlane(rank0[c]);
            row = 0, flushed = 0, carry = (T)0;
            if (!IDENT)
                rnext = nzrow[rank_base + lane];
        }
        st = nst;
        cur = nxt;
    }
}

/* blocks [0, fix_blocks): a wave per row that crosses chunk boundaries -- y[row] = carry[first] + ... + carry[last - 1] + the
 * piece k_hub left in y[row]: lane l adds carries l, l + 64, ... in that order, a fixed 64-lane tree, then the piece;
 * the blocks behind them: y = 0 for the rows without entries (native-impl.c:6,9) */
template <typename T>
__global__ void __launch_bounds__(256) k_hub_fix(const int *__restrict__ fix, int n_fix, int fix_blocks, const int *__restrict__ erow,
                                                 int n_empty, const T *__restrict__ carry, T *__restrict__ y)
{
    if ((int)blockIdx.x >= fix_blocks) {
        const int i = ((int)blockIdx.x - fix_blocks) * 256 + (int)threadIdx.x;
        if (i < n_empty)
            y[erow[i]] = (T)0;
        return;
    }
    const int k = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (k >= n_fix)
        return;
    const int row = fix[3 * k], cf = fix[3 * k + 1], cl = fix[3 * k + 2];
    T t = (T)0;
    for (int c = cf + lane; c < cl; c += 64)
        t = t + carry[c];
    t = group_sum<T, 64>(t);
    if (lane == 63)
        y[row] = t + y[row];
}
#pragma clang fp contract(fast)

template <typename T>
__global__ void __launch_bounds__(256) k_hub_pack(const int *__restrict__ hubcol, int n, const T *__restrict__ x, T *__restrict__ xh)
{
    const int s = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (s < n) {
        const int c = hubcol[s]; /* -1: the zero behind tier 0 and the pads up to tier 1 */
        xh[s] = c >= 0 ? x[c] : (T)0;
    }
}

/* ---------------------------------------------------------------- the builder */
__global__ void __launch_bounds__(256) k_hub_hist(const int *__restrict__ col, long long nnz, int *__restrict__ cnt)
{
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < nnz; j += (long long)gridDim.x * 256)
        atomicAdd(&cnt[col[j] - 1], 1);
}

__global__ void __launch_bounds__(256) k_hub_remap(const int *__restrict__ col, long long nnz, long long n_alloc,
                                                   const int *__restrict__ slot_of, unsigned int zero_slot, int K, int base1,
                                                   unsigned int *__restrict__ col2)
{
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n_alloc; j += (long long)gridDim.x * 256) {
        unsigned int v = HUB | zero_slot; /* the pads: 0 (the CSR copy's value pad) x 0 (the table's zero) */
        if (j < nnz) {
            const int c = col[j] - 1, s = slot_of[c];
            v = s < 0 ? (unsigned int)c : s < K ? (HUB | (unsigned int)s) : (TBL | (unsigned int)(base1 + s - K));
        }
        col2[j] = v;
    }
}

__global__ void __launch_bounds__(256) k_hub_ends(const int *__restrict__ rowptr, int rows, unsigned int *__restrict__ col2)
{
    const int r = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (r < rows && rowptr[r + 1] > rowptr[r])
        col2[rowptr[r + 1] - 1] |= END;
}

std::atomic<int> g_hub_min_cover{25}; /* per cent of the entries that must hit a hub for the automatic choice to keep the plan */
std::atomic<int> g_hub_table_bytes{SPMV_HUB_TABLE_BYTES}; /* tier 1: bytes of the dense table in memory (0: none) */
std::atomic<int> g_hub_lds_cap{1 << 30};                  /* tier 0: at most this many columns in the LDS table (tests: small, so
                                                             that small matrices reach tier 1 and x itself) */

} // namespace

#define HBCHK(call)                                                                 \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            rc = e_ == hipErrorOutOfMemory ? -3 : -1;                               \
            if (rc == -1)                                                           \
                spmv_set_error("HUB builder: %s: %s", #call, hipGetErrorString(e_)); \
            (void)hipGetLastError();                                                \
            goto out;                                                               \
        }                                                                           \
    } while (0)

extern "C" void spmv_hub_free(spmv_dev_hub *d)
{
    (void)hipFree(d->col2), (void)hipFree(d->hubcol), (void)hipFree(d->xh), (void)hipFree(d->rank0), (void)hipFree(d->nzrow);
    (void)hipFree(d->erow), (void)hipFree(d->fix), (void)hipFree(d->carry);
    memset(d, 0, sizeof *d);
}

extern "C" void spmvk_hub_set_min_cover(int per_cent)
{
    if (per_cent >= 0 && per_cent <= 100)
        g_hub_min_cover.store(per_cent, std::memory_order_relaxed);
}

extern "C" void spmvk_hub_set_lds_cap(int columns)
{
    g_hub_lds_cap.store(columns >= 0 ? columns : 1 << 30, std::memory_order_relaxed);
}

extern "C" void spmvk_hub_set_table_bytes(int bytes)
{
    if (bytes >= 0 && bytes <= (256 << 20))
        g_hub_table_bytes.store(bytes, std::memory_order_relaxed);
}

extern "C" int spmv_hub_build_device(const int *d_rowptr, const int *rowptr_host, const int *d_col, const void *d_val, int rows,
                                     int cols, int elem_size, int64_t nnz, int auto_choice, void *stream, spmv_dev_hub *out)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    memset(out, 0, sizeof *out);
    if (rows <= 0 || cols <= 0 || nnz <= 0 || nnz > 0x7fffffffll - 2 * SPMV_PAD_ELEMS || cols >= (1 << 29) || !rowptr_host)
        return -2;
    const int slots = spmv_hub_lds_slots(elem_size);
    const size_t n_alloc = (size_t)nnz + SPMV_PAD_ELEMS;
    const unsigned int grid = nnz < 256ll * 8192 ? (unsigned int)((nnz + 255) / 256) : 8192u;
    int *d_cnt = nullptr, *h_cnt = nullptr, *d_slot = nullptr, *h_hubcol = nullptr;
    int base1 = 0;
    size_t n_slots = 0;
    int hub_cus = 256;
    spmv_hub_plan g;
    spmv_dev_hub d;
    memset(&g, 0, sizeof g);
    memset(&d, 0, sizeof d);
    h_cnt = (int *)malloc((size_t)cols * sizeof(int));
    if (!h_cnt) {
        rc = -3;
        goto out;
    }
    /* ---- references per column, counted on the device (integers: the same in every build) */
    HBCHK(hipMalloc((void **)&d_cnt, (size_t)cols * sizeof(int)));
    HBCHK(hipMemsetAsync(d_cnt, 0, (size_t)cols * sizeof(int), st));
    hipLaunchKernelGGL(k_hub_hist, dim3(grid), dim3(256), 0, st, d_col, (long long)nnz, d_cnt);
    HBCHK(hipGetLastError());
    HBCHK(hipMemcpyAsync(h_cnt, d_cnt, (size_t)cols * sizeof(int), hipMemcpyDeviceToHost, st));
    HBCHK(hipStreamSynchronize(st));
    /* ---- the decisions (spmv_hub_plan.c) */
    {
        const int cap = g_hub_lds_cap.load(std::memory_order_relaxed);
        rc = spmv_hub_select(h_cnt, cols, cap < slots ? cap : slots, g_hub_table_bytes.load(std::memory_order_relaxed) / elem_size, 2, &g);
    }
    if (rc)
        goto out;
    base1 = (g.K + 1 + 15) / 16 * 16; /* tier 1 begins on a 64- / 128-byte boundary behind tier 0 and its zero */
    n_slots = (size_t)base1 + (size_t)g.K2;
    /* the automatic choice wants real hubs: the share of the entries that the LDS table alone takes.  The second tier does not
     * count -- 262 144 columns cover half of ANY matrix with half a million columns, and on uniformly scattered or banded columns
     * the plan then loses to the CSR kernels by 13-63 % (profiles/r04_dispatch_audit_before_the_criterion.log) */
    /* ... and hubs means: referenced four times as often as the average column that is referenced at all (the K most popular of
     * uniformly used columns cover K / columns of the entries and are no hubs, however few the columns) */
    if (auto_choice && (100 * g.covered < (int64_t)g_hub_min_cover.load(std::memory_order_relaxed) * nnz ||
                        (double)g.covered * (double)g.cols_used < 4.0 * (double)g.K * (double)nnz)) {
        rc = -2; /* no hubs to speak of: the CSR kernel gathers as well */
        goto out;
    }
    rc = spmv_hub_rows(rowptr_host, rowptr_host[0], rows, CH, &g);
    if (rc)
        goto out;
    /* ---- the remapped columns */
    HBCHK(hipMalloc((void **)&d_slot, (size_t)cols * sizeof(int)));
    HBCHK(hipMemcpyAsync(d_slot, g.slot_of, (size_t)cols * sizeof(int), hipMemcpyHostToDevice, st));
    HBCHK(hipMalloc((void **)&d.col2, n_alloc * sizeof(unsigned int)));
    hipLaunchKernelGGL(k_hub_remap, dim3(grid), dim3(256), 0, st, d_col, (long long)nnz, (long long)n_alloc, d_slot,
                       (unsigned int)g.K, g.K, base1, d.col2);
    HBCHK(hipGetLastError());
    hipLaunchKernelGGL(k_hub_ends, dim3((unsigned int)((rows + 255) / 256)), dim3(256), 0, st, d_rowptr, rows, d.col2);
    HBCHK(hipGetLastError());
    /* ---- the small arrays */
    /* the column behind every slot of xh: tier 0, -1 (the zero and the pads), tier 1 */
    h_hubcol = (int *)malloc((n_slots + 1) * sizeof(int));
    if (!h_hubcol) {
        rc = -3;
        goto out;
    }
    for (int s = 0; s < base1; ++s)
        h_hubcol[s] = s < g.K ? g.hubcol[s] : -1;
    if (g.K2)
        memcpy(h_hubcol + base1, g.hubcol + g.K, (size_t)g.K2 * sizeof(int));
    HBCHK(hipMalloc((void **)&d.hubcol, (n_slots + 1) * sizeof(int)));
    HBCHK(hipMemcpyAsync(d.hubcol, h_hubcol, n_slots * sizeof(int), hipMemcpyHostToDevice, st));
    HBCHK(hipMalloc(&d.xh, (n_slots + 16) * (size_t)elem_size));
    HBCHK(hipMemsetAsync(d.xh, 0, (n_slots + 16) * (size_t)elem_size, st));
    HBCHK(hipMalloc((void **)&d.rank0, ((size_t)g.n_chunks + 1) * sizeof(int)));
    HBCHK(hipMemcpyAsync(d.rank0, g.rank0, ((size_t)g.n_chunks + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    {
        /* a wave prefetches the rows of the NEXT 64 row ends: the list is padded so that it may run past the end */
        const size_t n_nz = (size_t)g.n_nzrows + 2 * HubGeom<float>::RING + 64;
        HBCHK(hipMalloc((void **)&d.nzrow, n_nz * sizeof(int)));
        HBCHK(hipMemsetAsync(d.nzrow, 0, n_nz * sizeof(int), st));
        if (g.n_nzrows)
            HBCHK(hipMemcpyAsync(d.nzrow, g.nzrow, (size_t)g.n_nzrows * sizeof(int), hipMemcpyHostToDevice, st));
    }
    HBCHK(hipMalloc((void **)&d.erow, ((size_t)g.n_empty + 1) * sizeof(int)));
    if (g.n_empty)
        HBCHK(hipMemcpyAsync(d.erow, g.erow, (size_t)g.n_empty * sizeof(int), hipMemcpyHostToDevice, st));
    HBCHK(hipMalloc((void **)&d.fix, ((size_t)g.n_fix * 3 + 1) * sizeof(int)));
    if (g.n_fix)
        HBCHK(hipMemcpyAsync(d.fix, g.fix, (size_t)g.n_fix * 3 * sizeof(int), hipMemcpyHostToDevice, st));
    HBCHK(hipMalloc(&d.carry, ((size_t)g.n_chunks + 1) * (size_t)elem_size));
    HBCHK(hipStreamSynchronize(st));
    {   /* one persistent workgroup per CU of the device the layout is built on */
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            hub_cus = cus;
    }
    d.val = d_val;
    d.elem = elem_size, d.rows = rows, d.cols = cols, d.K = g.K, d.K2 = g.K2, d.base1 = base1, d.slots = slots, d.CH = CH;
    d.n_chunks = g.n_chunks, d.covered2 = g.covered2;
    d.n_nzrows = g.n_nzrows, d.n_empty = g.n_empty, d.n_fix = g.n_fix, d.identity = g.identity;
    d.nnz = nnz, d.covered = g.covered;
    {
        int nwg = (g.n_chunks + WAVES - 1) / WAVES; /* a wave should find at least one chunk */
        if (nwg > hub_cus)
            nwg = hub_cus;
        d.n_wg = nwg < 1 ? 1 : nwg;
    }
    d.bytes = n_alloc * 4 + (n_slots + 1) * 4 + (n_slots + 16) * (size_t)elem_size + ((size_t)g.n_chunks + 1) * (4 + (size_t)elem_size) +
              ((size_t)g.n_nzrows + 2 * HubGeom<float>::RING + 64) * 4 + ((size_t)g.n_empty + 1) * 4 + ((size_t)g.n_fix * 3 + 1) * 4;
    *out = d;
    memset(&d, 0, sizeof d);
out:
    (void)hipFree(d_cnt), (void)hipFree(d_slot);
    free(h_cnt), free(h_hubcol);
    spmv_hub_plan_free(&g);
    if (rc)
        spmv_hub_free(&d);
    return rc;
}

namespace {
template <typename T> void launch_hub(const spmv_dev_hub *d, const void *x, void *y, hipStream_t st)
{
    using G = HubGeom<T>;
    const int n_slots = d->K2 > 0 ? d->base1 + d->K2 : d->K;
    if (n_slots > 0)
        hipLaunchKernelGGL(k_hub_pack<T>, dim3((unsigned int)((n_slots + 255) / 256)), dim3(256), 0, st, d->hubcol, n_slots, (const T *)x, (T *)d->xh);
    const int kpad = (d->K + 1 + G::E - 1) / G::E * G::E; /* the table and its zero, in 16-byte units */
    const int per = (d->n_wg + 7) / 8, grid = per * 8;
    if (d->identity)
        hipLaunchKernelGGL((k_hub<T, true>), dim3(grid), dim3(NT_WG), 0, st, d->col2, (const T *)d->val, (const T *)x, (const T *)d->xh,
                           d->rank0, d->nzrow, (T *)y, (T *)d->carry, d->n_chunks, kpad, d->n_wg, per);
    else
        hipLaunchKernelGGL((k_hub<T, false>), dim3(grid), dim3(NT_WG), 0, st, d->col2, (const T *)d->val, (const T *)x, (const T *)d->xh,
                           d->rank0, d->nzrow, (T *)y, (T *)d->carry, d->n_chunks, kpad, d->n_wg, per);
    if (d->n_fix > 0 || d->n_empty > 0) {
        const int fb = (d->n_fix + 3) / 4, eb = (d->n_empty + 255) / 256;
        hipLaunchKernelGGL(k_hub_fix<T>, dim3((unsigned int)(fb + eb)), dim3(256), 0, st, d->fix, d->n_fix, fb, d->erow, d->n_empty,
                           (const T *)d->carry, (T *)y);
    }
}
} // namespace

extern "C" int spmvk_hub(const spmv_dev_hub *d, const void *x, void *y, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (d->rows <= 0 || !d->col2) {
        spmv_set_error("HUB: no layout behind the plan");
        return (int)hipErrorInvalidValue;
    }
    if (d->slots != (d->elem == 8 ? HubGeom<double>::SLOTS : HubGeom<float>::SLOTS) || d->CH != CH) {
        spmv_set_error("HUB: layout planned for another kernel (%d slots, chunks of %d)", d->slots, d->CH);
        return (int)hipErrorInvalidValue;
    }
    if (d->elem == 8)
        launch_hub<double>(d, x, y, st);
    else
        launch_hub<float>(d, x, y, st);
    return (int)hipGetLastError();
}
