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

// max over 64 lanes of an unsigned 64-bit key, wave-uniform result
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
#define DQ_MAXSTEP(CTRL, MASK)                                                                             \
    {                                                                                                      \
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(k & 0xffffffffULL), CTRL, MASK, 0xf, false);   \
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(k >> 32), CTRL, MASK, 0xf, false);             \
        const unsigned long long o = ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;      \
        k = o > k ? o : k;                                                                                 \
    }
    DQ_MAXSTEP(0xB1, 0xf)
    DQ_MAXSTEP(0x4E, 0xf)
    DQ_MAXSTEP(0x141, 0xf)
    DQ_MAXSTEP(0x140, 0xf)
    DQ_MAXSTEP(0x142, 0xa)
    DQ_MAXSTEP(0x143, 0xc)
#undef DQ_MAXSTEP
    const int lo = __builtin_amdgcn_readlane((int)(k & 0xffffffffULL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(k >> 32), 63);
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}

}  // namespace dq
