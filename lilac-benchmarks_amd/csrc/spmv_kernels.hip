/*
 * csrc/spmv_kernels.hip -- hand-written CDNA4 (gfx950) kernels for CSR y = A*x.
 *
 * These replace the closed cusparse{D,S}csrmv_mp call of the reference GPU
 * backend (/root/reference/libspmv/gpu.c:270-282 and :350-362); the arithmetic
 * they must reproduce is /root/reference/libspmv/native-impl.c:1-25.
 *
 * The product is an HBM-bound integer/gather stream (AI = 0.166 flop/B for
 * FP64), so there is no MFMA anywhere: what matters is that a[] and colidx[]
 * are read exactly once with 16-byte-per-lane coalesced loads, that x[] stays
 * in L2, and that enough bytes are in flight per CU.
 *
 * Kernel family (256-thread workgroups = 4 wavefronts of 64, except k_tiled: 8 waves and
 * k_striped: 16 waves):
 *
 *   k_ordered   one lane per row, sequential, multiply then add with FP
 *               contraction off: bit-identical to the reference loop.
 *   k_vector    L lanes per row (L = 1,2,4,...,64; L = 64 is one wavefront per
 *               row), 16-byte aligned vector loads of a/colidx straight from
 *               global memory, DPP reduction inside the lane group.
 *   k_stream    row-segment kernel: a workgroup owns a run of whole rows with
 *               <= 2044 nonzeros, streams a/colidx coalesced regardless of where
 *               rows start, multiplies by the gathered x, parks the products in
 *               LDS, then lane groups (1..64 lanes, chosen per block) reduce each
 *               row out of LDS.
 *   k_long      rows longer than a row block: one workgroup per 2048-nonzero
 *               segment, partial sums combined in fixed order by k_long_finish
 *               (no float atomics: results are reproducible run to run).
 *   k_tiled     row blocks x column stripes with the stripe of x double-buffered in
 *               LDS; one lane per row adds in stored order (bit-exact on sorted
 *               columns).  Layout: spmv_tiled.c.
 *   k_striped   2-D decomposition: a workgroup = (group of column stripes, row block) keeps a
 *               stripe of x in LDS, streams packed (value, 16-bit column) entries -- with
 *               non-temporal loads when the stream cannot stay in the Infinity Cache anyway --
 *               and carries row sums in registers across the stripes of its group; k_group_sum
 *               adds the groups' partial sums in group order (a second, small launch).
 *               Layout: spmv_striped.c, built on the GPU by spmv_striped_gpu.hip.  The NPB
 *               class B/C/D kernel; no FP contraction: bit-identical to tests/striped_order.py.
 *
 * Column ids stay 1-based as the ABI delivers them; kernels index x through
 * (x - 1).  a/colidx are over-allocated by SPMV_PAD_ELEMS entries (a = 0,
 * colidx = 1) so that aligned vector loads may run past a row's end.
 */
#include <hip/hip_runtime.h>

#include "spmv_internal.h"
#include "spmv_wave.h"

namespace {

/* ---- tuning switches (set through spmvk_set_tuning; experiments only) ------- */
int g_tune_xcd = 1;     /* give each XCD a contiguous range of row blocks        */
int g_tune_nt = 0;      /* non-temporal loads for the streamed a / colidx        */
int g_tune_striped_nt = 0; /* k_striped's packed stream: 0 non-temporal loads when it exceeds the Infinity Cache, 1 always, 2 never */
int g_tune_tile_k = 0;  /* register sets of k_tiled (2,4,6,8; default 4); debug build: of k_striped too (2,3,4; default 2) */
#ifdef SPMV_HIP_DEBUG_BUILD
/* The instrumented twin of k_striped (template parameter DIAG) exists in the debug build only (make debug ->
 * libhip-spmv-debug.so, include/spmv_hip_debug.h): the shipped kernel carries no ablation tests and no stamp stores. */
int g_tune_abl = 0;     /* k_striped<DIAG> ablation bits (timing experiments only; results are wrong when set) */

/* diagnostic stamps of k_striped<DIAG> (ablation bit 64; tools/kbench.py --stamps): per workgroup the 100-MHz wall clock
 * at entry, after the first stripe of x is staged, and when wave 0 is done */
__device__ unsigned long long g_striped_stamps[1024 * 4];
#endif

/*
 * Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits vmcnt(0), which
 * would drain the matrix sub-windows a wave of k_tiled keeps in flight across stripe barriers;
 * here only LDS writes have to be visible to the other waves, so the fences name the local
 * address space and the wait they lower to is lgkmcnt(0).
 */
__device__ __forceinline__ void wg_barrier_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

/* Blocks b, b+8, b+16, ... share an XCD (and its L2): hand every XCD one
 * contiguous range of work items so neighbouring row blocks share cache lines
 * and x-locality inside one L2.  Launch with grid = 8 * ceil(n/8); ids >= n idle. */
__device__ __forceinline__ int xcd_contiguous(int b, int per_xcd) { return (b & 7) * per_xcd + (b >> 3); }

/* ================================ k_ordered ===================================== */
template <typename T>
__global__ void __launch_bounds__(SPMV_WG) k_ordered(const int *__restrict__ rowptr,
                                                     const int *__restrict__ colidx,
                                                     const T *__restrict__ val,
                                                     const T *__restrict__ xm1, T *__restrict__ y,
                                                     int rows)
{
#pragma clang fp contract(off)
    const int r = blockIdx.x * SPMV_WG + threadIdx.x;
    if (r >= rows)
        return;
    const int s = rowptr[r], e = rowptr[r + 1];
    T acc = (T)0;
    for (int j = s; j < e; ++j) {
        T p = val[j] * xm1[colidx[j]];
        acc = acc + p;
    }
    y[r] = acc;
}

/* ================================ k_vector ====================================== */
template <typename T, int L, bool LIST, bool NT>
__global__ void __launch_bounds__(SPMV_WG) k_vector(const int *__restrict__ rowptr,
                                                    const int *__restrict__ colidx,
                                                    const T *__restrict__ val,
                                                    const T *__restrict__ xm1, T *__restrict__ y,
                                                    const int *__restrict__ row_list, int nrows,
                                                    int per_xcd)
{
    using VT = typename Vec16<T>::val_t;
    using IT = typename Vec16<T>::idx_t;
    constexpr int E = Vec16<T>::E;
    constexpr int RPB = SPMV_WG / L; /* rows per workgroup */

    const int tid = threadIdx.x;
    const int wg = per_xcd > 0 ? xcd_contiguous(blockIdx.x, per_xcd) : (int)blockIdx.x;
    const int g = wg * RPB + tid / L;
    const int lane = tid % L;

    T acc0 = (T)0, acc1 = (T)0;
    int row = 0;
    if (g < nrows) {
        row = LIST ? row_list[g] : g;
        const int s = rowptr[row], e = rowptr[row + 1];
        int j = (s & ~(E - 1)) + lane * E;
        for (; j < e; j += 2 * L * E) {
            const int j2 = j + L * E;
            const bool two = j2 < e;
            IT c0 = ld<IT, NT>(reinterpret_cast<const IT *>(colidx + j));
            VT v0 = ld<VT, NT>(reinterpret_cast<const VT *>(val + j));
            IT c1 = c0;
            VT v1 = v0;
            if (two) {
                c1 = ld<IT, NT>(reinterpret_cast<const IT *>(colidx + j2));
                v1 = ld<VT, NT>(reinterpret_cast<const VT *>(val + j2));
            }
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int jj = j + i;
                if (jj >= s && jj < e)
                    acc0 = fmadd(v0[i], xm1[c0[i]], acc0);
            }
            if (two) {
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const int jj = j2 + i;
                    if (jj < e)
                        acc1 = fmadd(v1[i], xm1[c1[i]], acc1);
                }
            }
        }
    }
    T acc = group_sum<T, L>(acc0 + acc1);
    if (g < nrows && lane == L - 1)
        y[row] = acc;
}

/* ================================ k_stream ====================================== */
/* DIAG: the debug build's timing-only twin (tuning key 6; results are wrong when a bit is set): 1 = no phase 2 (what do
 * the loads, the gathers and the LDS parking cost alone), 2 = every gather goes to the first KiB of x (what do the
 * gather MISSES cost), 4 = no row pointers either, 8 = the gathers of R-MAT's popular columns -- those whose 0-based index has
 * at most (mask >> 8) bits set -- go to the first KiB (what would an LDS-resident table of the hub columns leave).  The
 * shipped instantiation (DIAG = false) contains none of it. */
template <typename T, bool NT, bool DIAG>
__global__ void __launch_bounds__(SPMV_WG) k_stream(const int *__restrict__ blk,
                                                    const int *__restrict__ rowptr,
                                                    const int *__restrict__ colidx,
                                                    const T *__restrict__ val,
                                                    const T *__restrict__ xm1, T *__restrict__ y,
                                                    int nblk, int per_xcd, int abl_arg)
{
    const int abl = DIAG ? abl_arg : 0;
    using VT = typename Vec16<T>::val_t;
    using IT = typename Vec16<T>::idx_t;
    constexpr int E = Vec16<T>::E;
    constexpr int CAP = SPMV_STREAM_CAP;
    constexpr int ROUNDS = CAP / (SPMV_WG * E);

    __shared__ __attribute__((aligned(16))) T s_prod[CAP];
    __shared__ int s_rp[SPMV_STREAM_ROWCAP + 1];

    const int b = per_xcd > 0 ? xcd_contiguous(blockIdx.x, per_xcd) : (int)blockIdx.x;
    if (b >= nblk)
        return;
    const int tid = threadIdx.x;
    const i32x2 d0 = *reinterpret_cast<const i32x2 *>(blk + 2 * b);
    const i32x2 d1 = *reinterpret_cast<const i32x2 *>(blk + 2 * b + 2);
    const int r0 = d0[0], s = d0[1], r1 = d1[0], e = d1[1];
    const int nrows = r1 - r0;
    const int base = s & ~(E - 1);
    if (e - s > CAP - 4) /* a long row's block (planner cap = CAP - 4): k_long owns it */
        return;

    /* ---- phase 1: stream a / colidx, gather x, park products in LDS ---- */
    /* All loads of the block are issued before anything waits on them, with no
     * branch in between: a lane whose slot lies past the block's end re-reads
     * its own round-0 slot instead (an L1 hit; at worst SPMV_WG*E entries past a
     * tiny block's end, inside the SPMV_PAD_ELEMS padding).  Entries before s /
     * after e belong to neighbouring rows or the padding: their column ids are
     * valid and their products are never read back. */
    IT c[ROUNDS];
    VT v[ROUNDS];
    const int j0 = base + tid * E;
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int j = j0 + k * SPMV_WG * E;
        const int jl = j < e ? j : j0;
        c[k] = ld<IT, NT>(reinterpret_cast<const IT *>(colidx + jl));
        v[k] = ld<VT, NT>(reinterpret_cast<const VT *>(val + jl));
    }
    if (!(abl & 4))
        for (int i = tid; i <= nrows; i += SPMV_WG)
            s_rp[i] = rowptr[r0 + i] - base;
    T xg[ROUNDS][E];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k)
#pragma unroll
        for (int i = 0; i < E; ++i)
            xg[k][i] = xm1[((abl & 2) || ((abl & 8) && __builtin_popcount((unsigned)c[k][i] - 1u) <= (abl >> 8))) ? 1 + (c[k][i] & 127)
                                                                                                                    : c[k][i]];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        VT p;
#pragma unroll
        for (int i = 0; i < E; ++i)
            p[i] = v[k][i] * xg[k][i];
        *reinterpret_cast<VT *>(&s_prod[(k * SPMV_WG + tid) * E]) = p;
    }
    __syncthreads();
    if (abl & 1) { /* timing only */
        if (tid < nrows)
            y[r0 + tid] = s_prod[tid];
        return;
    }

    /* ---- phase 2: lane groups reduce whole rows out of LDS ---- */
    /* G lanes per row, block-uniform: as many as fill the workgroup in one pass,
     * but no more than the average row offers */
    const int n = e - s;
    int G = 64;
    while (G > 1 && (G * nrows > SPMV_WG || G > 2 * ((n + nrows - 1) / (nrows > 0 ? nrows : 1))))
        G >>= 1;

    /* A row much longer than its block's average would keep one small lane group busy long after
     * the others are done (R-MAT blocks: 130 rows of ~15 entries and one of 2000).  Lane groups
     * leave such rows alone; afterwards every wave gives all 64 lanes to each of the ones it finds
     * among its share of the rows. */
    const int far = G >= 64 ? CAP : (32 * G > 64 ? 32 * G : 64);
    int skipped = 0;

#define SPMV_REDUCE_ROWS(GG)                                                         \
    {                                                                                \
        constexpr int NG = SPMV_WG / GG;                                             \
        const int grp = tid / GG, gl = tid % GG;                                     \
        /* rr is uniform inside a lane group, so every DPP step below only ever   \
         * combines lanes that are active together */                              \
        for (int rr = grp; rr < nrows; rr += NG) {                                   \
            const int ps = s_rp[rr], pe = s_rp[rr + 1];                              \
            if (pe - ps > far) {                                                     \
                skipped = 1;                                                         \
                continue;                                                            \
            }                                                                        \
            T acc = (T)0;                                                            \
            for (int k = ps + gl; k < pe; k += GG)                                   \
                acc += s_prod[k];                                                    \
            acc = group_sum<T, GG>(acc);                                             \
            if (gl == GG - 1)                                                        \
                y[r0 + rr] = acc;                                                    \
        }                                                                            \
    }
    switch (G) {
    case 64: SPMV_REDUCE_ROWS(64) break;
    case 32: SPMV_REDUCE_ROWS(32) break;
    case 16: SPMV_REDUCE_ROWS(16) break;
    case 8: SPMV_REDUCE_ROWS(8) break;
    case 4: SPMV_REDUCE_ROWS(4) break;
    case 2: SPMV_REDUCE_ROWS(2) break;
    default: SPMV_REDUCE_ROWS(1) break;
    }
    if (!__syncthreads_or(skipped)) /* the usual case: nothing was left behind */
        return;
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int base_r = 0; base_r < nrows; base_r += SPMV_WG) {
            const int rr = base_r + wave * 64 + lane;
            const bool mine = rr < nrows && (s_rp[rr < nrows ? rr + 1 : 0] - s_rp[rr < nrows ? rr : 0]) > far;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(mine);
            while (todo) {
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int row = base_r + wave * 64 + l;
                const int ps = s_rp[row], pe = s_rp[row + 1];
                T acc = (T)0;
                for (int k = ps + lane; k < pe; k += 64)
                    acc += s_prod[k];
                acc = group_sum<T, 64>(acc);
                if (lane == 63)
                    y[r0 + row] = acc;
            }
        }
    }
#undef SPMV_REDUCE_ROWS
}


/* ================================ k_tiled ======================================= */
/*
 * TILED plan (layout: spmv_tiled.c).  One workgroup = WPB wavefronts = one row block; it walks
 * the column stripes in order, keeping the stripe of x in LDS (s_x, 64 KiB), so the per-nonzero
 * gather is an LDS bank access instead of an L2 line.  Inside a stripe every wave is on its
 * own: it streams its chunk of packed (value, 16-bit local column) pairs with coalesced
 * 16-byte loads, 256 entries per pass, parks the products in its private LDS window (s_p) and
 * then each lane adds up, front to back, the products of the one row it owns.  The accumulator
 * of a row lives in a register of its lane across all stripes, y is written once, nothing is
 * atomic, and the order of the additions is the stored order.
 */
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

/* One sub-window of packed entries held in registers: 4 entries per lane, the four 16-bit
 * stripe-local columns kept packed in two dwords. */
template <typename T, int EPL> struct TileRegs {
    static_assert(EPL == 4, "four entries per lane");
    T v[4];
    unsigned int c01, c23;
    __device__ __forceinline__ unsigned int col(int k) const
    {
        const unsigned int pk = k < 2 ? c01 : c23;
        return (k & 1) ? (pk >> 16) : (pk & 0xffffu);
    }
};

/* entries [q + 4*lane, +4) of the packed stream; q < 2^31 - padding, so 32-bit offsets */
template <typename T, int EPL>
__device__ __forceinline__ void tile_load(TileRegs<T, EPL> &r, const T *__restrict__ pval,
                                          const unsigned short *__restrict__ pcol, int q, int lane)
{
    using VT = typename Vec16<T>::val_t;
    constexpr int E = Vec16<T>::E;
    const unsigned int e = (unsigned int)q + (unsigned int)lane * EPL;
    const u32x2 cc = *reinterpret_cast<const u32x2 *>(pcol + e);
    r.c01 = cc[0];
    r.c23 = cc[1];
#pragma unroll
    for (int k = 0; k < EPL; k += E) {
        const VT vv = *reinterpret_cast<const VT *>(pval + e + k);
#pragma unroll
        for (int i = 0; i < E; ++i)
            r.v[k + i] = vv[i];
    }
}

template <typename T, int WPB, int R, int XT_BYTES>
__global__ void __launch_bounds__(WPB * 64, 4) k_tiled(const int *__restrict__ wptr,
                                                       const unsigned short *__restrict__ seg,
                                                       const unsigned short *__restrict__ pcol,
                                                       const T *__restrict__ pval,
                                                       const T *__restrict__ x, T *__restrict__ y,
                                                       int rows, int cols, int S, int RW)
{
    using VT = typename Vec16<T>::val_t;
    constexpr int E = Vec16<T>::E;
    constexpr int W = XT_BYTES / (int)sizeof(T); /* columns per stripe */
    constexpr int EPL = 4;                       /* entries per lane per sub-window */
    constexpr int SUB = 64 * EPL;                /* entries per sub-window = products staged per pass */
    constexpr int NT = WPB * 64;
    constexpr int XV = W / (NT * E);             /* 16-byte x loads per thread per stripe */
    static_assert(XV * NT * E == W, "stripe must be a whole number of 16-byte loads per thread");
    static_assert((R + 2) * SUB <= SPMV_TILE_PAD, "look-ahead must stay inside the padding");
    __shared__ __attribute__((aligned(16))) T s_x[2][W];
    __shared__ __attribute__((aligned(16))) T s_p[WPB][SUB];

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int g = blockIdx.x * WPB + w;
    const long long ch0 = (long long)g * S;
    T acc = (T)0;

    /*
     * Stripe s+1 of x travels global -> registers while stripe s is being consumed and is
     * written to the idle LDS buffer when the wave leaves stripe s: one barrier per stripe.  No
     * branch around any load (a branch makes hipcc wait vmcnt(0) per load): a lane whose 16
     * bytes would cross the end of x reads x[0..] instead -- its columns are never referenced --
     * and the <16-byte tail of the last stripe is patched in by the thread that owns that slot.
     * x must be 16-byte aligned (the launcher checks).
     */
    VT xr[XV];
    auto x_fetch = [&](int s) {
        const int c0 = s * W;
        const int nx = (cols - c0) < W ? (cols - c0) : W;
        const T *xs = x + c0;
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const int i = (k * NT + tid) * E;
            const T *src = (i + E <= nx) ? xs + i : x;
            xr[k] = *reinterpret_cast<const VT *>(src);
        }
    };
    auto x_store = [&](int buf) {
#pragma unroll
        for (int k = 0; k < XV; ++k)
            *reinterpret_cast<VT *>(&s_x[buf][(k * NT + tid) * E]) = xr[k];
    };
    const int tail = cols % E;
    const int tail_pos = (cols - tail) - (S - 1) * W; /* slot of the first tail entry in its stripe */
    const bool tail_mine = tail != 0 && ((tail_pos / E) % NT) == tid;
    auto tail_fix = [&](int s) {
        if (tail_mine && s == S - 1)
            for (int j = 0; j < tail; ++j)
                s_x[s & 1][tail_pos + j] = x[cols - tail + j];
    };

    /*
     * The wave's packed entries are one linear stream starting at wptr[ch0].  It is read in
     * sub-windows of 256 entries at fixed stream positions, whatever the chunk boundaries;
     * sub-window j lives in register set j % R and is reloaded with sub-window j + R the moment
     * it has been consumed, with no register copies, so every load is full.  The loop nest is
     * sub-window-major; a wave moves on to the next stripe (x_store, barrier) whenever the chunk
     * it is in ends inside the current sub-window, and every wave of the block meets exactly S
     * barriers.  The barrier orders LDS only (wg_barrier_lds), so the look-ahead survives it.
     */
    const int P0 = __builtin_amdgcn_readfirstlane(wptr[ch0]);
    TileRegs<T, EPL> set[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        tile_load<T, EPL>(set[r], pval, pcol, P0 + r * SUB, lane);

    /* row-segment offsets of the wave's rows inside the coming chunk, fetched one stripe ahead */
    const unsigned short *sg = seg + ch0 * (RW + 1);
    const int sl = lane < RW ? lane : RW; /* idle lanes read the total twice: an empty segment */
    const int sl1 = lane < RW ? lane + 1 : RW;
    int ns_s = sg[sl], ns_e = sg[sl1];
    x_fetch(0);
    x_store(0);
    tail_fix(0);
    wg_barrier_lds();

    int s = 0;        /* stripe the wave is in */
    int q = P0;       /* stream position of its chunk */
    int q_e, n;       /* end of the chunk, entries in it */
    int abs_s, abs_e; /* stream positions of this lane's row inside the chunk */
    const T *xt;
    auto begin_stripe = [&]() {
        xt = s_x[s & 1];
        n = __builtin_amdgcn_readlane(ns_e, RW - 1);
        abs_s = q + ns_s;
        abs_e = q + ns_e;
        q_e = q + n;
        if (s + 1 < S) {
            x_fetch(s + 1);
            sg += RW + 1;
            ns_s = sg[sl];
            ns_e = sg[sl1];
        }
    };
    auto end_stripe = [&]() {
        q += (n + 7) & ~7;
        if (s + 1 < S) {
            x_store((s + 1) & 1); /* that buffer was last read in stripe s-1, which every wave
                                     left before the previous barrier */
            tail_fix(s + 1);
        }
        wg_barrier_lds();
        ++s;
    };
    begin_stripe();

    for (int jj = 0;; jj += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int sb = P0 + (jj + r) * SUB; /* wave-uniform stream position of set[r] */
            const TileRegs<T, EPL> cur = set[r]; /* copy out: one counted wait here, none inside the loop */
            tile_load<T, EPL>(set[r], pval, pcol, sb + R * SUB, lane);
            for (;;) {
                if (sb < q_e && sb + SUB > q) {
                    /* every lane multiplies (entries outside [q, q_e) give products that are never
                     * read; the & keeps their LDS address inside the stripe) */
                    T p[EPL];
#pragma unroll
                    for (int k = 0; k < EPL; ++k)
                        p[k] = cur.v[k] * xt[cur.col(k) & (W - 1)];
#pragma unroll
                    for (int k = 0; k < EPL; ++k)
                        s_p[w][lane * EPL + k] = p[k];
                    __builtin_amdgcn_wave_barrier(); /* LDS ops of one wave complete in issue order */
                    const int lo = (abs_s > sb ? abs_s : sb) - sb;
                    const int hi = (abs_e < sb + SUB ? abs_e : sb + SUB) - sb;
                    int k = lo;
                    for (; k + 4 <= hi; k += 4) {
                        const T p0 = s_p[w][k], p1 = s_p[w][k + 1], p2 = s_p[w][k + 2], p3 = s_p[w][k + 3];
                        acc = acc + p0;
                        acc = acc + p1;
                        acc = acc + p2;
                        acc = acc + p3;
                    }
                    for (; k < hi; ++k)
                        acc = acc + s_p[w][k];
                    __builtin_amdgcn_wave_barrier();
                }
                if (q_e > sb + SUB)
                    break; /* the chunk goes on in the next sub-window */
                end_stripe();
                if (s == S)
                    goto done;
                begin_stripe();
            }
        }
    }
done:
    const int row = g * RW + lane;
    if (lane < RW && row < rows)
        y[row] = acc;
}


/* ================================ k_striped ===================================== */
/*
 * STRIPED plan (layout: spmv_striped.c).  Workgroup = (stripe group g, row block rb of that group).  It walks the
 * stripes of its group: copy the stripe of x into LDS (one staging, two LDS-only barriers), then
 * every wave is on its own until the stripe ends, streaming the tile of ITS rows -- packed (value,
 * flagged 16-bit column) entries -- through a ring of register sets reloaded in place; the wave's
 * tiles follow each other in memory, so the ring simply runs on across stripe changes.  Per
 * 256-entry sub-window: multiply by x from LDS, add each lane's four entries up to the row-end
 * flags, segmented scan over the wave (DPP) for the rows that span lanes; rows are numbered by
 * counting flags (ballot + mbcnt): no row pointers are read.  A finished (row, stripe) sum is parked
 * in the wave's LDS ring and leaves it 64 rows at a time INTO REGISTERS: accumulator k of lane l
 * belongs to row 64 k + l of the wave's range and lives across all stripes of the group.  After the
 * group the accumulators are y (NG = 1) or the group's partial sums, which k_group_sum adds in group
 * order.  Nothing is atomic; the order of every addition is fixed by the layout.
 */
/* One sub-window of the STRIPED stream in registers: EPL consecutive entries per lane.  Inside
 * every block of 64*EPL entries the values are stored lane-interleaved in 16-byte units (see
 * spmv_striped.c), so each of the EPL*sizeof(T)/16 value loads is a dense 1-KiB wave access. */
template <typename T, int EPL> struct StripeRegs {
    static constexpr int E = Vec16<T>::E;
    static constexpr int NV = EPL / E; /* 16-byte value loads per lane */
    typename Vec16<T>::val_t v[NV];
    unsigned int c[EPL / 2];           /* two 16-bit columns per dword */
    __device__ __forceinline__ T val(int k) const { return v[k / E][k % E]; }
    __device__ __forceinline__ unsigned int col(int k) const
    {
        return (k & 1) ? (c[k >> 1] >> 16) : (c[k >> 1] & 0xffffu);
    }
};

/* NT: non-temporal loads.  A packed stream larger than the 256-MB Infinity Cache is read once per product and never found
 * there again; loading it without allocating leaves the cache to x and the partial sums and the memory system to stream:
 * class C 73.7 -> 71.0 us, a class D row block 172.0 -> 156.6, half of class D 605 -> 559 (profiles/r03_striped_nt_ab.log).
 * A stream that FITS the cache (class B: 138 MB) is served from it on every product after the first, and non-temporal
 * loads forfeit that: 28.1 -> 31.2 us.  The launcher decides by the stream's size. */
template <typename T, int EPL, bool NT>
__device__ __forceinline__ void stripe_load(StripeRegs<T, EPL> &r, const T *__restrict__ pval,
                                            const unsigned short *__restrict__ pcol, int sb, int lane)
{
    using VT = typename Vec16<T>::val_t;
    constexpr int E = Vec16<T>::E;
    const unsigned int base = (unsigned int)sb;
    static_assert(EPL == 4, "one 8-byte column load per lane");
    const u32x2 cc = ld<u32x2, NT>(reinterpret_cast<const u32x2 *>(pcol + base + (unsigned int)lane * 4u));
    r.c[0] = cc[0], r.c[1] = cc[1];
#pragma unroll
    for (int j = 0; j < StripeRegs<T, EPL>::NV; ++j)
        r.v[j] = ld<VT, NT>(reinterpret_cast<const VT *>(pval + base + (unsigned int)(j * 64 * E) + (unsigned int)lane * E));
}

/* acc[k] += v in the lanes where pred holds.  k is wave-uniform (made so with readfirstlane), which lets the
 * compiler address the register array through s_set_gpr_idx_on: two indexed moves out, one add, two moves back,
 * instead of sixteen copies of the add behind a branch tree (9000 lines of ISA: more than the instruction
 * cache) or an array in scratch memory. */
template <typename T, int NACC> __device__ __forceinline__ void acc_add(T (&acc)[NACC], unsigned int k, T v, bool pred)
{
    const int ki = __builtin_amdgcn_readfirstlane((int)k) & (NACC - 1);
    const T cur = acc[ki];
    acc[ki] = pred ? cur + v : cur;
}

/* No floating-point contraction in k_striped and k_group_sum: with every product rounded before it is added, the result
 * is a function of the layout alone, addition by addition, and tests/striped_order.py can restate it in numpy and compare
 * the GPU's bits with it (the vector pipe is not what bounds the kernel: class C 73.2 us with, 73.9 without the pragma,
 * a class D row block 168.0 / 168.0; profiles/r03_striped_fp_contract_ab.log) */
#pragma clang fp contract(off)
template <typename T, int WPB, int R, int XT_BYTES, int EPL, int NACC, bool DIAG, bool NTLOAD>
__global__ void __launch_bounds__(WPB * 64, 4)
k_striped(const int *__restrict__ tptr, const int *__restrict__ rrow, const int *__restrict__ wgd,
          const unsigned short *__restrict__ pcol, const T *__restrict__ pval, const T *__restrict__ x,
          T *__restrict__ y, T *__restrict__ partial, const T *__restrict__ dotv, T *__restrict__ dot_part, int rows,
          int cols, int NG, int GM, int wcols, int nwg, int per_xcd, int abl_arg)
{
    using VT = typename Vec16<T>::val_t;
    constexpr int E = Vec16<T>::E;
    const int abl = DIAG ? abl_arg : 0; /* the shipped instantiation (DIAG = false) folds every test of it away */
    constexpr int W = XT_BYTES / (int)sizeof(T); /* elements of the LDS buffer; W - 1 is the zero column */
    constexpr int SUB = 64 * EPL;
    constexpr int NT = WPB * 64;
    constexpr int XV = W / (NT * E);
    static_assert(XV * NT * E == W, "stripe must be a whole number of 16-byte loads per thread");
    static_assert((R + 1) * SUB <= SPMV_TILE_PAD, "look-ahead must stay inside the padding");
    static_assert(R >= 2 && R <= 4, "ring depth");
    constexpr int RING = 256; /* row sums staged per wave before they are taken into the accumulators */
    __shared__ __attribute__((aligned(16))) T s_x[W];
    __shared__ __attribute__((aligned(16))) T s_out[WPB][RING];

    const int q = per_xcd ? xcd_contiguous((int)blockIdx.x, per_xcd) : (int)blockIdx.x;
    if (q >= nwg)
        return; /* the whole workgroup */
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
#ifdef SPMV_HIP_DEBUG_BUILD
    if (DIAG && (abl & 64) && tid == 0 && q < 1024)
        g_striped_stamps[4 * q] = __builtin_amdgcn_s_memrealtime();
#endif
    const int s0 = __builtin_amdgcn_readfirstlane(wgd[4 * q]), s1 = __builtin_amdgcn_readfirstlane(wgd[4 * q + 1]);
    const int g = __builtin_amdgcn_readfirstlane(wgd[4 * q + 2]);
    const int u = q * WPB + w; /* wave range */
    const int row0 = __builtin_amdgcn_readfirstlane(rrow[2 * u]);
    const unsigned int nr = (unsigned int)(__builtin_amdgcn_readfirstlane(rrow[2 * u + 1]) - row0);
    const int *__restrict__ tp = tptr + (size_t)u * (size_t)(GM + 1);

    T acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
        acc[k] = (T)0;

    /* the first R sub-windows are on their way while x is staged */
    int P = __builtin_amdgcn_readfirstlane(tp[0]);
    StripeRegs<T, EPL> set[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        stripe_load<T, EPL, NTLOAD>(set[r], pval, pcol, P + r * SUB, lane);

    /* per-stripe state of the wave */
    unsigned int row = 0, flushed = 0; /* local row numbers: next row to end, first row still in the ring */
    T carry = (T)0;                    /* open sum of the row that spans the sub-window boundary */

    /* rows [a, b) leave the ring for the accumulators; a is a multiple of 64 (it always is: see below), b - a <= 64:
     * lane l takes row a + l into accumulator a / 64 */
    auto take = [&](const unsigned int a, const unsigned int b) __attribute__((always_inline)) {
        const unsigned int r = a + (unsigned int)lane;
        if (abl & 4)
            return;
        const T v = s_out[w][r & (RING - 1)];
        acc_add<T, NACC>(acc, a >> 6, v, r < b);
    };

    /*
     * One sub-window.  The main loop below calls this R times per trip with no exit in between, so
     * the R register sets stay a static ring (no copies, and the compiler can count the loads in
     * flight: it waits for exactly the set it consumes).
     */
    auto sub_window = [&](StripeRegs<T, EPL> &regs, const int sb) __attribute__((always_inline)) {
        T p[EPL];
        int f[EPL];
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const unsigned int c = regs.col(k);
            p[k] = regs.val(k) * s_x[c & (W - 1)];
            f[k] = (int)(c >> 15);
        }
        stripe_load<T, EPL, NTLOAD>(regs, pval, pcol, sb + R * SUB, lane); /* may run into the wave's next tile: same stream */
        if (abl & 32) { /* timing only: what a layout with one row per lane would leave of a sub-window (loads, LDS gathers, EPL adds) */
            T t = (T)0;
#pragma unroll
            for (int k = 0; k < EPL; ++k)
                t = t + p[k];
            acc[0] = acc[0] + t;
            return;
        }
        /* lane-local: the sum after the lane's last row end, and whether it has a row end */
        T tailv = (T)0;
        int has = 0;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            tailv = tailv + p[k]; /* the flagged entry itself still belongs to the closing row */
            if (f[k])
                tailv = (T)0;
            has |= f[k];
        }
        /* inclusive segmented scan of (tailv, has) over the wave */
        T sv = tailv;
        int sf = has;
        if (!(abl & 16)) {
        segscan_step<T, 0x111, 0xf>(sv, sf); /* row_shr:1 */
        segscan_step<T, 0x112, 0xf>(sv, sf); /* row_shr:2 */
        segscan_step<T, 0x114, 0xf>(sv, sf); /* row_shr:4 */
        segscan_step<T, 0x118, 0xf>(sv, sf); /* row_shr:8 */
        segscan_step<T, 0x142, 0xa>(sv, sf); /* row_bcast:15 */
        segscan_step<T, 0x143, 0xc>(sv, sf); /* row_bcast:31 */
        }
        if (!sf)
            sv = sv + carry; /* no row end at or before this lane: the previous sub-window's open sum */
        T cin = dpp_mov<0x138, 0xf>(sv); /* wave_shr:1: what the lane below leaves open */
        if (lane == 0)
            cin = carry;
        carry = __shfl(sv, 63, 64);

        /* row numbers: count the row ends in lower lanes */
        unsigned int before = 0, nflags = 0;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const unsigned long long m = __ballot(f[k] != 0);
            before += __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32),
                                                __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0));
            nflags += (unsigned int)__builtin_popcountll(m);
        }
        /*
         * Park the sums of the rows that end in this lane in the wave's ring (slot = row number mod 256), then
         * take whole groups of 64 rows out of it.  The ring holds < 64 rows from earlier sub-windows and this
         * one adds up to 256: rows beyond the ring's reach (a sub-window crowded with one-entry rows) wait for
         * a second pass, after the first has emptied the ring down to < 64 rows again -- `flushed` stays a
         * multiple of 64, so every take() is one indexed accumulator.
         */
        const unsigned int flushed0 = flushed, row_new = row + nflags;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            T run = cin;
            unsigned int idx = row + before;
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                run = run + p[k];
                if (f[k]) {
                    if ((idx - flushed0 >= (unsigned int)RING) == (pass == 1) && !(abl & 8))
                        s_out[w][idx & (RING - 1)] = run;
                    ++idx;
                    run = (T)0;
                }
            }
            const unsigned int present = pass == 0 && row_new - flushed0 > (unsigned int)RING ? flushed0 + RING : row_new;
            __builtin_amdgcn_wave_barrier();
            while (present - flushed >= 64u) {
                take(flushed, flushed + 64u);
                flushed += 64u;
            }
            __builtin_amdgcn_wave_barrier();
            if (row_new - flushed0 <= (unsigned int)RING) /* wave-uniform: nearly always */
                break;
        }
        row = row_new;
    };

    for (int si = 0; s0 + si < s1; ++si) {
        /* ---- the stripe of x into LDS ---- */
        {
            const int c0 = (s0 + si) * wcols; /* wcols <= W - 16/sizeof(T) columns per stripe (spmv_striped_geometry) */
            const int nx = (cols - c0) < wcols ? (cols - c0) : wcols;
            const T *xs = x + c0;
            VT xr[XV];
#pragma unroll
            for (int k = 0; k < XV; ++k) {
                const int i = (k * NT + tid) * E;
                const T *src = (i + E <= nx) ? xs + i : x; /* no branch around a load; see k_tiled */
                xr[k] = *reinterpret_cast<const VT *>(src);
            }
            if (si > 0)
                wg_barrier_lds(); /* every wave is done with the previous stripe */
#pragma unroll
            for (int k = 0; k < XV; ++k)
                *reinterpret_cast<VT *>(&s_x[(k * NT + tid) * E]) = xr[k];
            const int tail = nx % E; /* <16-byte tail of the last stripe: patched by the slot's owner */
            if (tail != 0 && (((nx - tail) / E) % NT) == tid)
                for (int j = 0; j < tail; ++j)
                    s_x[nx - tail + j] = xs[nx - tail + j];
            if (tid == NT - 1) /* the thread that wrote the buffer's last 16 bytes: fillers and pads multiply with 0 */
                s_x[W - 1] = (T)0;
            wg_barrier_lds();
#ifdef SPMV_HIP_DEBUG_BUILD
            if (DIAG && (abl & 64) && si == 0 && tid == 0 && q < 1024)
                g_striped_stamps[4 * q + 1] = __builtin_amdgcn_s_memrealtime();
#endif
        }
        /* ---- the wave's tile ---- */
        const int Pend = __builtin_amdgcn_readfirstlane(tp[si + 1]);
        const int nsub = (Pend - P) / SUB;
        row = 0, flushed = 0, carry = (T)0;
        int j = 0;
        for (; j + R <= nsub; j += R) {
            if (!(abl & 128)) {
                /*
                 * The four waves that share a SIMD (w, w+4, w+8, w+12) take turns at every priority, a turn = 4
                 * sub-windows.  Left alone, the arbiter favours the older wave: wave 0 of a class C workgroup was
                 * done after 51 us and the last wave after 65 (clock stamps, tools/kbench.py --stamps), and a CU
                 * whose waves retire one by one has ever fewer bytes in flight for its last quarter.  Worth 0.4-1.1 %
                 * (12 interleaved A/B rounds: class C 75.8 against 76.4 us, B 28.7 / 29.0, a class D row block
                 * 177.6 / 178.3); tuning key 6 bit 128 switches it off.
                 */
                switch (((j >> 2) + (w >> 2)) & 3) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                sub_window(set[r], P + (j + r) * SUB);
        }
        /* the last nsub % R sub-windows, then turn the ring so that set[0] is again the next sub-window */
#pragma unroll
        for (int k = 1; k < R; ++k)
            if (nsub - j == k) { /* wave-uniform */
#pragma unroll
                for (int r = 0; r < k; ++r)
                    sub_window(set[r], P + (j + r) * SUB);
                StripeRegs<T, EPL> tmp[R];
#pragma unroll
                for (int i = 0; i < R; ++i)
                    tmp[i] = set[(i + k) % R];
#pragma unroll
                for (int i = 0; i < R; ++i)
                    set[i] = tmp[i];
            }
        P = Pend;
        __builtin_amdgcn_wave_barrier();
        take(flushed, row); /* the last < 64 rows of the tile */
        __builtin_amdgcn_wave_barrier();
    }

#ifdef SPMV_HIP_DEBUG_BUILD
    if (DIAG && (abl & 64) && lane == 0 && q < 1024) { /* wave 0: its own end; wave WPB - 1: the last wave's (a proxy for the workgroup's) */
        if (w == 0)
            g_striped_stamps[4 * q + 2] = __builtin_amdgcn_s_memrealtime();
        if (w == WPB - 1)
            g_striped_stamps[4 * q + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    /* ---- the group is done: acc[k] of lane l is the sum of local row 64 k + l over the group's stripes ---- */
    const bool folded = dotv != nullptr; /* kernel-uniform */
    T dot = (T)0;
    if (NG == 1) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            const unsigned int r = 64u * k + (unsigned int)lane;
            if (r < nr) {
                y[row0 + r] = acc[k];
                if (folded)
                    dot = fmadd(acc[k], dotv[row0 + r], dot);
            }
        }
    } else { /* the group's partial sums; k_group_sum adds the groups in order */
        T *__restrict__ out = partial + (size_t)g * (size_t)rows + row0;
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            const unsigned int r = 64u * k + (unsigned int)lane;
            if (r < nr)
                out[r] = acc[k];
        }
        return;
    }
    if (folded) { /* dot_part[u] = sum over the wave range of y[r] * dotv[r]: per lane in row order, then a butterfly */
        for (int o = 32; o; o >>= 1)
            dot += __shfl_xor(dot, o, 64);
        if (lane == 0)
            dot_part[u] = dot;
    }
}

/* y[r] = sum of the NG group partials of row r, in group order.  With `dotv` the kernel also leaves
 * dot_part[block] = sum over the block's rows of y[r] * dotv[r] (fixed order: wave butterfly, then the
 * four waves in order) -- the p.q of a CG step without another pass over y (spmv_cg.hip).
 *
 * Why a second launch and not the last workgroup to arrive (tried in round 2, profiles/r02_kbench_fold_ablation.log):
 * the waves of a launch finish together, so whoever adds up a row range does it on the critical path, after
 * store -> drain -> arrival counter -> acquire -> loads; one wave per range pulling NG * rows-per-range values
 * took 19 us on class C and 45 us on a class D row block, against 5-8 us for this kernel on the whole chip. */
template <typename T>
__global__ void __launch_bounds__(SPMV_WG) k_group_sum(const T *__restrict__ partial, T *__restrict__ y,
                                                       int rows, int NG, const T *__restrict__ dotv,
                                                       T *__restrict__ dot_part)
{
    const int r = blockIdx.x * SPMV_WG + threadIdx.x;
    T acc = (T)0;
    if (r < rows) {
        const T *p = partial + r;
        int g = 0;
        for (; g + 8 <= NG; g += 8) { /* eight loads in flight, added in group order */
            T v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = p[(size_t)(g + k) * rows];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                acc = acc + v[k];
        }
        for (; g < NG; ++g)
            acc = acc + p[(size_t)g * rows];
        y[r] = acc;
    }
    if (dotv) { /* kernel-uniform */
        __shared__ T s_w[SPMV_WG / 64];
        T d = r < rows ? acc * dotv[r] : (T)0;
        for (int o = 32; o; o >>= 1)
            d += __shfl_xor(d, o, 64);
        if ((threadIdx.x & 63) == 0)
            s_w[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0)
            dot_part[blockIdx.x] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
    }
}

#pragma clang fp contract(fast)

/* ================================ k_long ======================================== */
template <typename T>
__global__ void __launch_bounds__(SPMV_WG) k_long(const int *__restrict__ seg,
                                                  const int *__restrict__ colidx,
                                                  const T *__restrict__ val,
                                                  const T *__restrict__ xm1, T *__restrict__ y,
                                                  T *__restrict__ partials)
{
    using VT = typename Vec16<T>::val_t;
    using IT = typename Vec16<T>::idx_t;
    constexpr int E = Vec16<T>::E;
    __shared__ T s_w[SPMV_WG / 64];

    const i32x4 d = *reinterpret_cast<const i32x4 *>(seg + 4 * blockIdx.x);
    const int row = d[0], sb = d[1], se = d[2], slot = d[3];
    const int tid = threadIdx.x;
    /* A segment is a handful of 16-byte loads per thread: all of them (and then all the gathers)
     * are issued before the first product, so a workgroup pays the memory latency about twice,
     * not once per pass.  Slots beyond the segment re-read its first vector and contribute 0. */
    constexpr int U = SPMV_LONG_SEG / (SPMV_WG * E) + 1; /* +1: sb need not be 16-byte aligned */
    const int j0 = (sb & ~(E - 1)) + tid * E;
    IT c[U];
    VT v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = j0 + u * SPMV_WG * E;
        const int jl = j < se ? j : j0;
        c[u] = *reinterpret_cast<const IT *>(colidx + jl);
        v[u] = *reinterpret_cast<const VT *>(val + jl);
    }
    T xv[U][E];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < E; ++i)
            xv[u][i] = xm1[c[u][i]];
    T acc = (T)0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = j0 + u * SPMV_WG * E;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int jj = j + i;
            if (jj >= sb && jj < se)
                acc = fmadd(v[u][i], xv[u][i], acc);
        }
    }
    acc = group_sum<T, 64>(acc);
    if ((tid & 63) == 63)
        s_w[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        T t = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
        if (slot < 0)
            y[row] = t;
        else
            partials[slot] = t;
    }
}

/* One wavefront per long row: lane l adds the row's segment sums l, l+64, ... in that order, then a
 * fixed 64-lane tree.  (A dense row of 10^8 nonzeros has 49 000 segment sums; one thread adding them
 * one after the other would take milliseconds.) */
template <typename T>
__global__ void __launch_bounds__(64) k_long_finish(const int *__restrict__ lrow,
                                                    const T *__restrict__ partials,
                                                    T *__restrict__ y, int nlrow)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= nlrow)
        return;
    const int row = lrow[3 * k], p0 = lrow[3 * k + 1], p1 = lrow[3 * k + 2];
    T t = (T)0;
    for (int p = p0 + lane; p < p1; p += 64)
        t += partials[p];
    t = group_sum<T, 64>(t);
    if (lane == 63)
        y[row] = t;
}

/* ================================ analysis ====================================== */
__global__ void __launch_bounds__(SPMV_WG) k_colrange(const int *__restrict__ colidx, long long nnz,
                                                       int *__restrict__ out /* [0] max, [1] min */)
{
    int m = 0, n = 0x7fffffff;
    for (long long j = (long long)blockIdx.x * SPMV_WG + threadIdx.x; j < nnz;
         j += (long long)gridDim.x * SPMV_WG) {
        int c = colidx[j];
        m = c > m ? c : m;
        n = c < n ? c : n;
    }
    for (int o = 32; o; o >>= 1) {
        int t = __shfl_xor(m, o, 64);
        m = t > m ? t : m;
        t = __shfl_xor(n, o, 64);
        n = t < n ? t : n;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(out, m);
        atomicMin(out + 1, n);
    }
}

/* word copy; src may be pinned host memory mapped into the device's address space (zero-copy) */
__global__ void __launch_bounds__(SPMV_WG) k_copy_words(unsigned int *__restrict__ dst,
                                                        const unsigned int *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * SPMV_WG + threadIdx.x; i < n; i += (size_t)gridDim.x * SPMV_WG)
        dst[i] = src[i];
}

__global__ void __launch_bounds__(SPMV_WG) k_rebase(int *__restrict__ rowptr, int n, int base)
{
    const int i = blockIdx.x * SPMV_WG + threadIdx.x;
    if (i < n)
        rowptr[i] -= base;
}

inline int launch_status() { return (int)hipGetLastError(); }

template <typename T> const T *xm1_of(const void *x) { return reinterpret_cast<const T *>(x) - 1; }

template <typename T, int L>
int launch_vector(const spmv_dev_csr *m, const int *row_list, int nrows, const void *x, void *y,
                  hipStream_t st)
{
    constexpr int RPB = SPMV_WG / L;
    const int nwg = (nrows + RPB - 1) / RPB;
    if (nwg <= 0)
        return 0;
    const int per = g_tune_xcd ? (nwg + 7) / 8 : 0;
    const int grid = per ? per * 8 : nwg;
    const T *val = reinterpret_cast<const T *>(m->val);
    T *yy = reinterpret_cast<T *>(y);
#define SPMV_LV(LIST, NT)                                                                        \
    hipLaunchKernelGGL((k_vector<T, L, LIST, NT>), dim3(grid), dim3(SPMV_WG), 0, st, m->rowptr,  \
                       m->colidx, val, xm1_of<T>(x), yy, row_list, nrows, per)
    if (row_list) {
        if (g_tune_nt)
            SPMV_LV(true, true);
        else
            SPMV_LV(true, false);
    } else {
        if (g_tune_nt)
            SPMV_LV(false, true);
        else
            SPMV_LV(false, false);
    }
#undef SPMV_LV
    return launch_status();
}

template <typename T>
int dispatch_vector(const spmv_dev_csr *m, int lanes, const int *row_list, int nrows, const void *x,
                    void *y, hipStream_t st)
{
    switch (lanes) {
    case 1: return launch_vector<T, 1>(m, row_list, nrows, x, y, st);
    case 2: return launch_vector<T, 2>(m, row_list, nrows, x, y, st);
    case 4: return launch_vector<T, 4>(m, row_list, nrows, x, y, st);
    case 8: return launch_vector<T, 8>(m, row_list, nrows, x, y, st);
    case 16: return launch_vector<T, 16>(m, row_list, nrows, x, y, st);
    case 32: return launch_vector<T, 32>(m, row_list, nrows, x, y, st);
    case 64: return launch_vector<T, 64>(m, row_list, nrows, x, y, st);
    default: return (int)hipErrorInvalidValue;
    }
}

/* the two instantiated shapes of k_striped: 16 waves over a 128-KiB stripe (one workgroup per CU) and 8 waves over a
 * 64-KiB stripe (two per CU, which run out of step: one streams while the other stages x or waits at its barrier) */
template <typename T, int WPB, int XB, int R, bool DIAG, int NACC = SPMV_STRIPE_NACC>
void launch_striped(const spmv_dev_striped *t, const void *x, void *y, hipStream_t st, const void *dotv,
                           void *dot_part, int grid, int nwg, int per, int abl)
{
    /* non-temporal loads of the packed stream when it cannot stay in the Infinity Cache anyway (see stripe_load) */
    const long long stream = (long long)t->n_packed * (long long)(sizeof(T) + 2);
    const bool nt = g_tune_striped_nt == 1 || (g_tune_striped_nt == 0 && stream > SPMV_STRIPE_NT_BYTES);
    if (nt)
        hipLaunchKernelGGL((k_striped<T, WPB, R, XB, SPMV_STRIPE_EPL, NACC, DIAG, true>), dim3(grid), dim3(WPB * 64), 0,
                           st, t->tptr, t->rrow, t->wgd, t->col, (const T *)t->val, (const T *)x, (T *)y, (T *)t->partial,
                           (const T *)dotv, (T *)dot_part, t->rows, t->cols, t->NG, t->GM, t->W, nwg, per, abl);
    else
        hipLaunchKernelGGL((k_striped<T, WPB, R, XB, SPMV_STRIPE_EPL, NACC, DIAG, false>), dim3(grid), dim3(WPB * 64), 0,
                           st, t->tptr, t->rrow, t->wgd, t->col, (const T *)t->val, (const T *)x, (T *)y, (T *)t->partial,
                           (const T *)dotv, (T *)dot_part, t->rows, t->cols, t->NG, t->GM, t->W, nwg, per, abl);
}

template <typename T, int WPB, int XB>
void launch_striped_shape(const spmv_dev_striped *t, const void *x, void *y, hipStream_t st, const void *dotv,
                                 void *dot_part, int grid, int nwg, int per)
{
#ifdef SPMV_HIP_DEBUG_BUILD
    /* experiments of round 4 (profiles/r04_classC_where_the_step_goes.md): fewer accumulators for layouts whose wave ranges are
     * short, which frees registers for a deeper ring -- clean instantiations, no ablation code in them.  Tuning key 5:
     * 12 = 2 sets, 4 accumulators (the control); 14 = 4 sets, 4 accumulators; 13 = 3 sets, 8 accumulators */
    if (g_tune_abl == 0 && g_tune_tile_k >= 12 && g_tune_tile_k <= 14) {
        if (g_tune_tile_k == 12 && t->wave_rows <= 256)
            return launch_striped<T, WPB, XB, 2, false, 4>(t, x, y, st, dotv, dot_part, grid, nwg, per, 0);
        if (g_tune_tile_k == 14 && t->wave_rows <= 256)
            return launch_striped<T, WPB, XB, 4, false, 4>(t, x, y, st, dotv, dot_part, grid, nwg, per, 0);
        if (g_tune_tile_k == 13 && t->wave_rows <= 512)
            return launch_striped<T, WPB, XB, 3, false, 8>(t, x, y, st, dotv, dot_part, grid, nwg, per, 0);
    }
    if (g_tune_abl != 0 || g_tune_tile_k == 3 || g_tune_tile_k == 4) { /* the instrumented twin, on request only */
        if (g_tune_tile_k == 4)
            launch_striped<T, WPB, XB, 4, true>(t, x, y, st, dotv, dot_part, grid, nwg, per, g_tune_abl);
        else if (g_tune_tile_k == 3)
            launch_striped<T, WPB, XB, 3, true>(t, x, y, st, dotv, dot_part, grid, nwg, per, g_tune_abl);
        else
            launch_striped<T, WPB, XB, 2, true>(t, x, y, st, dotv, dot_part, grid, nwg, per, g_tune_abl);
        return;
    }
#endif
    launch_striped<T, WPB, XB, 2, false>(t, x, y, st, dotv, dot_part, grid, nwg, per, 0);
}

} // namespace

extern "C" {

int spmvk_get_tuning_xcd(void) { return g_tune_xcd; }

void spmvk_set_tuning(int key, int value)
{
    if (key == 0)
        g_tune_xcd = value;
    else if (key == 1)
        g_tune_nt = value;
    else if (key == 5 && value >= 0 && value <= 14)
        g_tune_tile_k = value;
    else if (key == 20 && value >= 0 && value <= 2)
        g_tune_striped_nt = value;
#ifdef SPMV_HIP_DEBUG_BUILD
    else if (key == 6)
        g_tune_abl = value;
#endif
}

#ifdef SPMV_HIP_DEBUG_BUILD
/* copies the diagnostic stamps of the last k_striped<DIAG> launch with ablation bit 64 (debug build only) */
int spmvk_striped_stamps(unsigned long long *out, int n)
{
    if (n > 1024 * 4)
        n = 1024 * 4;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_striped_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

int spmvk_ordered(const spmv_dev_csr *m, const void *x, void *y, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int grid = (m->rows + SPMV_WG - 1) / SPMV_WG;
    if (grid <= 0)
        return 0;
    if (m->elem_size == 8)
        hipLaunchKernelGGL(k_ordered<double>, dim3(grid), dim3(SPMV_WG), 0, st, m->rowptr, m->colidx,
                           (const double *)m->val, xm1_of<double>(x), (double *)y, m->rows);
    else
        hipLaunchKernelGGL(k_ordered<float>, dim3(grid), dim3(SPMV_WG), 0, st, m->rowptr, m->colidx,
                           (const float *)m->val, xm1_of<float>(x), (float *)y, m->rows);
    return launch_status();
}

int spmvk_vector(const spmv_dev_csr *m, int lanes, const int *row_list, int nrows, const void *x,
                 void *y, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    return m->elem_size == 8 ? dispatch_vector<double>(m, lanes, row_list, nrows, x, y, st)
                             : dispatch_vector<float>(m, lanes, row_list, nrows, x, y, st);
}

int spmvk_stream(const spmv_dev_csr *m, const int *blk, int nblk, const void *x, void *y, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (nblk <= 0)
        return 0;
    const int per = g_tune_xcd ? (nblk + 7) / 8 : 0;
    const int grid = per ? per * 8 : nblk;
#define SPMV_LS(T, NT)                                                                          \
    hipLaunchKernelGGL((k_stream<T, NT, false>), dim3(grid), dim3(SPMV_WG), 0, st, blk, m->rowptr, \
                       m->colidx, (const T *)m->val, xm1_of<T>(x), (T *)y, nblk, per, 0)
#ifdef SPMV_HIP_DEBUG_BUILD
    if (g_tune_abl != 0) { /* the instrumented twin, on request only */
        if (m->elem_size == 8)
            hipLaunchKernelGGL((k_stream<double, false, true>), dim3(grid), dim3(SPMV_WG), 0, st, blk, m->rowptr, m->colidx,
                               (const double *)m->val, xm1_of<double>(x), (double *)y, nblk, per, g_tune_abl);
        else
            hipLaunchKernelGGL((k_stream<float, false, true>), dim3(grid), dim3(SPMV_WG), 0, st, blk, m->rowptr, m->colidx,
                               (const float *)m->val, xm1_of<float>(x), (float *)y, nblk, per, g_tune_abl);
        return launch_status();
    }
#endif
    if (m->elem_size == 8) {
        if (g_tune_nt)
            SPMV_LS(double, true);
        else
            SPMV_LS(double, false);
    } else {
        if (g_tune_nt)
            SPMV_LS(float, true);
        else
            SPMV_LS(float, false);
    }
#undef SPMV_LS
    return launch_status();
}

int spmvk_long(const spmv_dev_csr *m, const int *seg, int nseg, const int *lrow, int nlrow,
               void *partials, const void *x, void *y, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (nseg <= 0)
        return 0;
    if (m->elem_size == 8) {
        hipLaunchKernelGGL(k_long<double>, dim3(nseg), dim3(SPMV_WG), 0, st, seg, m->colidx,
                           (const double *)m->val, xm1_of<double>(x), (double *)y, (double *)partials);
        if (nlrow > 0)
            hipLaunchKernelGGL(k_long_finish<double>, dim3(nlrow), dim3(64), 0, st, lrow,
                               (const double *)partials, (double *)y, nlrow);
    } else {
        hipLaunchKernelGGL(k_long<float>, dim3(nseg), dim3(SPMV_WG), 0, st, seg, m->colidx,
                           (const float *)m->val, xm1_of<float>(x), (float *)y, (float *)partials);
        if (nlrow > 0)
            hipLaunchKernelGGL(k_long_finish<float>, dim3(nlrow), dim3(64), 0, st, lrow,
                               (const float *)partials, (float *)y, nlrow);
    }
    return launch_status();
}

int spmvk_tiled(const spmv_dev_tiled *t, const void *x, void *y, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (t->B <= 0)
        return 0;
    if (t->WPB != SPMV_TILE_WPB || t->W * t->elem != SPMV_TILE_LDS_BYTES)
        return (int)hipErrorInvalidValue;
#define SPMV_LT(T, R)                                                                                  \
    hipLaunchKernelGGL((k_tiled<T, SPMV_TILE_WPB, R, SPMV_TILE_LDS_BYTES>), dim3(t->B),                \
                       dim3(SPMV_TILE_WPB * 64), 0, st, t->wptr, t->seg, t->col,                       \
                       (const T *)t->val, (const T *)x, (T *)y, t->rows, t->cols, t->S, t->RW)
    /* the shipped library holds the one shape the plan uses (4 register sets); 2, 6 and 8 -- the last two spill -- are tuning
     * aids of the debug build (tuning key 5; profiles/r01_kbench11_tiled_ablation.log) */
#ifdef SPMV_HIP_DEBUG_BUILD
#define SPMV_LTK(T)                                                                                    \
    do {                                                                                               \
        if (g_tune_tile_k == 2)                                                                        \
            SPMV_LT(T, 2);                                                                             \
        else if (g_tune_tile_k == 6)                                                                   \
            SPMV_LT(T, 6);                                                                             \
        else if (g_tune_tile_k == 8)                                                                   \
            SPMV_LT(T, 8);                                                                             \
        else                                                                                           \
            SPMV_LT(T, 4);                                                                             \
    } while (0)
#else
#define SPMV_LTK(T) SPMV_LT(T, 4)
#endif
    if (t->elem == 8)
        SPMV_LTK(double);
    else
        SPMV_LTK(float);
#undef SPMV_LTK
#undef SPMV_LT
    return launch_status();
}

int spmvk_striped(const spmv_dev_striped *t, const void *x, void *y, void *stream, const void *dotv,
                  void *dot_part)
{
    hipStream_t st = (hipStream_t)stream;
    if (t->rows <= 0)
        return 0;
    const bool wide = t->WPB == SPMV_STRIPE_WPB && t->xbytes == SPMV_STRIPE_XBYTES;
    const bool half = t->WPB == SPMV_STRIPE2_WPB && t->xbytes == SPMV_STRIPE2_XBYTES;
    if ((!wide && !half) || t->EPL != SPMV_STRIPE_EPL || t->wave_rows > SPMV_STRIPE_WAVE_ROWS || (t->NG > 1 && !t->partial))
        return (int)hipErrorInvalidValue;
    const int nwg = t->n_wg;
    const int per = g_tune_xcd ? (nwg + 7) / 8 : 0; /* workgroups of one stripe group share an XCD's L2 for x */
    const int grid = per ? per * 8 : nwg;
    if (t->elem == 8) {
        if (wide)
            launch_striped_shape<double, SPMV_STRIPE_WPB, SPMV_STRIPE_XBYTES>(t, x, y, st, dotv, dot_part, grid, nwg, per);
        else
            launch_striped_shape<double, SPMV_STRIPE2_WPB, SPMV_STRIPE2_XBYTES>(t, x, y, st, dotv, dot_part, grid, nwg, per);
    } else {
        if (wide)
            launch_striped_shape<float, SPMV_STRIPE_WPB, SPMV_STRIPE_XBYTES>(t, x, y, st, dotv, dot_part, grid, nwg, per);
        else
            launch_striped_shape<float, SPMV_STRIPE2_WPB, SPMV_STRIPE2_XBYTES>(t, x, y, st, dotv, dot_part, grid, nwg, per);
    }
    if (t->NG > 1) {
        const int gs = (t->rows + SPMV_WG - 1) / SPMV_WG;
        if (t->elem == 8)
            hipLaunchKernelGGL(k_group_sum<double>, dim3(gs), dim3(SPMV_WG), 0, st, (const double *)t->partial,
                               (double *)y, t->rows, t->NG, (const double *)dotv, (double *)dot_part);
        else
            hipLaunchKernelGGL(k_group_sum<float>, dim3(gs), dim3(SPMV_WG), 0, st, (const float *)t->partial,
                               (float *)y, t->rows, t->NG, (const float *)dotv, (float *)dot_part);
    }
    return launch_status();
}

int spmvk_colrange(const int *colidx, int64_t nnz, int *d_out, void *stream)
{
    if (nnz <= 0)
        return 0;
    long long want = (nnz + SPMV_WG - 1) / SPMV_WG;
    int grid = (int)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(k_colrange, dim3(grid), dim3(SPMV_WG), 0, (hipStream_t)stream, colidx,
                       (long long)nnz, d_out);
    return launch_status();
}

int spmvk_copy_words(void *dst, const void *src, size_t bytes, void *stream)
{
    const size_t n = bytes / 4;
    if (n == 0)
        return 0;
    size_t grid = (n + SPMV_WG - 1) / SPMV_WG;
    if (grid > 1024)
        grid = 1024;
    hipLaunchKernelGGL(k_copy_words, dim3((unsigned int)grid), dim3(SPMV_WG), 0, (hipStream_t)stream,
                       (unsigned int *)dst, (const unsigned int *)src, n);
    return launch_status();
}

int spmvk_rebase(int *rowptr, int n_plus_1, int base, void *stream)
{
    if (n_plus_1 <= 0 || base == 0)
        return 0;
    hipLaunchKernelGGL(k_rebase, dim3((n_plus_1 + SPMV_WG - 1) / SPMV_WG), dim3(SPMV_WG), 0,
                       (hipStream_t)stream, rowptr, n_plus_1, base);
    return launch_status();
}

} // extern "C"
