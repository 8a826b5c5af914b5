/*
 * csrc/spmv_wave.h -- wave-level building blocks shared by the kernel files (spmv_kernels.hip, spmv_blocked.hip):
 * 16-byte vector types, DPP cross-lane moves, the lane-group sum with a fixed order of additions, one step of the
 * segmented wave scan.  Device code only; everything sits in an anonymous namespace (one copy per translation unit).
 */
#ifndef SPMV_WAVE_H
#define SPMV_WAVE_H

#include <hip/hip_runtime.h>

namespace {

/* ---- 16-byte vector types --------------------------------------------------- */
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Vec16;
template <> struct Vec16<double> {
    using val_t = f64x2;
    using idx_t = i32x2;
    static constexpr int E = 2;
};
template <> struct Vec16<float> {
    using val_t = f32x4;
    using idx_t = i32x4;
    static constexpr int E = 4;
};

template <typename V, bool NT> __device__ __forceinline__ V ld(const V *p)
{
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}


__device__ __forceinline__ double fmadd(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fmadd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

/* ---- DPP cross-lane adds ------------------------------------------------------ */
/* One DPP move of a 32- or 64-bit value; lanes whose source is disabled by
 * row_mask read 0 (old = 0, bound_ctrl off). */
/* A shift over all four rows (row_shr:n = 0x111..0x11f, wave_shr:1 = 0x138, row_mask 0xf) leaves no lane without a
 * defined result when bound_ctrl makes the lanes without a source read 0: then the move needs no "old" operand and the
 * compiler no v_mov 0 in front of it (two per 64-bit move, ten per sub-window of k_striped). */
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_mov_bits(int v)
{
    if constexpr (ROW_MASK == 0xf && ((CTRL >= 0x111 && CTRL <= 0x11f) || CTRL == 0x138))
        return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
    else
        return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, dpp_mov_bits<CTRL, ROW_MASK>(__builtin_bit_cast(int, v)));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_mov(double v)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = dpp_mov_bits<CTRL, ROW_MASK>(lo);
    hi = dpp_mov_bits<CTRL, ROW_MASK>(hi);
    long long r = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, r);
}

/*
 * Sum over aligned groups of G consecutive lanes (G = 1..64, power of two).
 * After the call the LAST lane of every group holds the group's sum (for
 * G <= 16 every lane does).  The order of the additions is fixed by G alone,
 * so a row's result does not depend on which lanes it landed on.
 */
template <typename T, int G> __device__ __forceinline__ T group_sum(T v)
{
    if constexpr (G >= 2)
        v += dpp_mov<0xB1, 0xf>(v); /* quad_perm [1,0,3,2] */
    if constexpr (G >= 4)
        v += dpp_mov<0x4E, 0xf>(v); /* quad_perm [2,3,0,1] */
    if constexpr (G >= 8)
        v += dpp_mov<0x141, 0xf>(v); /* row_half_mirror     */
    if constexpr (G >= 16)
        v += dpp_mov<0x140, 0xf>(v); /* row_mirror          */
    if constexpr (G >= 32)
        v += dpp_mov<0x142, 0xa>(v); /* row_bcast:15 -> rows 1,3 */
    if constexpr (G >= 64)
        v += dpp_mov<0x143, 0xc>(v); /* row_bcast:31 -> rows 2,3 */
    return v;
}

template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_mov(int v)
{
    return dpp_mov_bits<CTRL, ROW_MASK>(v);
}

/* one step of the inclusive segmented scan: (v, f) <- (prev o cur) with prev fetched by DPP */
template <typename T, int CTRL, int ROW_MASK> __device__ __forceinline__ void segscan_step(T &v, int &f)
{
    const T pv = dpp_mov<CTRL, ROW_MASK>(v);
    const int pf = dpp_mov<CTRL, ROW_MASK>(f);
    v = f ? v : pv + v;
    f |= pf;
}


} // namespace

#endif
