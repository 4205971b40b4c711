// wave.h -- wave64 cross-lane helpers for gfx950 (DPP / readlane, no LDS traffic).
#pragma once
#include <hip/hip_runtime.h>

namespace dq {

// one DPP move of a double (two 32-bit halves with the same control word);
// lanes that the row mask disables, or whose source lane is invalid, receive 0.0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double readlane_f64(double x, int lane /* wave-uniform */) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes, result broadcast to every lane (wave-uniform).
// quad_perm / row mirrors give every lane of a 16-lane row its row sum; row_bcast15/31
// (gfx9 DPP) fold the four rows into lane 63; v_readlane broadcasts it.
__device__ __forceinline__ double wave_sum(double x) {
    x += dpp_mov_f64<0xB1, 0xf>(x);        // quad_perm [1,0,3,2]
    x += dpp_mov_f64<0x4E, 0xf>(x);        // quad_perm [2,3,0,1]
    x += dpp_mov_f64<0x141, 0xf>(x);       // row_half_mirror
    x += dpp_mov_f64<0x140, 0xf>(x);       // row_mirror
    x += dpp_mov_f64<0x142, 0xa>(x);       // row_bcast:15 -> rows 1,3
    x += dpp_mov_f64<0x143, 0xc>(x);       // row_bcast:31 -> rows 2,3
    return readlane_f64(x, 63);
}

// Sum over each 16-lane DPP row; every lane of a row receives its row's sum.
__device__ __forceinline__ double row16_sum(double x) {
    x += dpp_mov_f64<0xB1, 0xf>(x);
    x += dpp_mov_f64<0x4E, 0xf>(x);
    x += dpp_mov_f64<0x141, 0xf>(x);
    x += dpp_mov_f64<0x140, 0xf>(x);
    return x;
}

// butterfly sum with LDS-crossbar shuffles (reference implementation for tests)
__device__ __forceinline__ double wave_sum_shfl(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// max over the 64 lanes of a 32-bit unsigned key (wave-uniform result): one v_max_u32 with a DPP operand per stage
// (the compiler folds update_dpp + max into v_max_u32_dpp; lanes a row mask disables keep their own value)
__device__ __forceinline__ unsigned wave_max_u32(unsigned k) {
#define DQ_MAX32(CTRL, MASK) { const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, CTRL, MASK, 0xf, false); k = o > k ? o : k; }
    DQ_MAX32(0xB1, 0xf)        // quad_perm [1,0,3,2]
    DQ_MAX32(0x4E, 0xf)        // quad_perm [2,3,0,1]
    DQ_MAX32(0x141, 0xf)       // row_half_mirror
    DQ_MAX32(0x140, 0xf)       // row_mirror
    DQ_MAX32(0x142, 0xa)       // row_bcast:15
    DQ_MAX32(0x143, 0xc)       // row_bcast:31
#undef DQ_MAX32
    return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
}

// max over 64 lanes of an unsigned 64-bit key, wave-uniform result: the high words first, then the low words of the lanes that
// tie on the high word -- the lexicographic maximum, i.e. exactly the 64-bit maximum, in 12 single-instruction DPP stages (a
// 64-bit compare-and-select chain needs five instructions per stage)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned m1 = wave_max_u32(hi);
    const unsigned m2 = wave_max_u32(hi == m1 ? lo : 0u);
    return ((unsigned long long)m1 << 32) | m2;
}

}  // namespace dq
