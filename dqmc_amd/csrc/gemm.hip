// gemm.hip -- batched n x n x n fp64 GEMM on the gfx950 matrix cores.
//
// Replaces every dense `*` of the reference's hot path (source/dqmc.cpp:76,82,
// 102,130,185; source/stablelinalg.cpp:61,66,73,78,85,91,149,150,157) and fuses
// the diagonal scalings the reference applies as separate passes
// (stablelinalg::diag_mul_mat / mat_mul_diag, source/stablelinalg.cpp:9-21;
// DQMC::calculate_B / calculate_invB, source/dqmc.cpp:78-86):
//
//     C = diag(rs) * ( op(A) * diag(ks) * B ) * diag(cs)   [+ C]
//
// Tiling: one wave owns one 16x16 tile of C and walks K with
// v_mfma_f64_16x16x4_f64; a 256-thread block is a 2x2 arrangement of waves
// (32x32 of C).  The product is formed transposed (MFMA-A <- B^T, MFMA-B <-
// op(A)^T) so that the accumulator's lane index runs along the ROWS of the
// column-major C and every store instruction writes 128-byte segments.
// Within a 16-deep K block lane group kk (= lane>>4) takes k = k0+4*kk+s at
// step s, so the B operand (and A^T) is read as 32 contiguous bytes per lane.
// Operands are read straight from L2: at the n <= 576 sizes of this code a
// tile's panels are a few tens of KB and every matrix is L2/MALL resident.
#include "common.h"
#include <cstdlib>

namespace dq {

using d4 = __attribute__((ext_vector_type(4))) double;

template <bool TRANSA, bool GUARD>
__global__ __launch_bounds__(256) void gemm_kernel(GemmDesc g, int tiles_per_dim) {
    const int chain = blockIdx.y;
    const int n = g.n;
    const double* __restrict__ A = g.A.at(chain);
    const double* __restrict__ B = g.B.at(chain);
    double* __restrict__ C = g.C.at(chain);
    const double* __restrict__ ks = g.ks.p ? g.ks.at(chain) : nullptr;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int bt_i = blockIdx.x % tiles_per_dim, bt_j = blockIdx.x / tiles_per_dim;
    const int i0 = bt_i * 32 + (wave & 1) * 16;
    const int j0 = bt_j * 32 + (wave >> 1) * 16;
    if (GUARD && (i0 >= n || j0 >= n)) return;

    const int r = lane & 15, kk = lane >> 4;
    const int ia = i0 + r;          // row of op(A) this lane feeds (MFMA-B operand column)
    const int jb = j0 + r;          // column of B this lane feeds (MFMA-A operand row)
    const bool ia_ok = !GUARD || ia < n, jb_ok = !GUARD || jb < n;

    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
    for (int k0 = 0; k0 < n; k0 += 16) {
        const int kb = k0 + 4 * kk;
        double av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = kb + s;
            const bool k_ok = !GUARD || k < n;
            bv[s] = (k_ok && jb_ok) ? B[k + (long)n * jb] : 0.0;
            double a = 0.0;
            if (k_ok && ia_ok) {
                a = TRANSA ? A[k + (long)n * ia] : A[ia + (long)n * k];
                if (ks) a *= ks[k];
            }
            av[s] = a;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[s], av[s], acc, 0, 0, 0);
    }
    // lane holds C[i0 + r][j0 + kk + 4*reg]
    if (ia_ok) {
        const double rsv = g.rs.p ? g.rs.at(chain)[ia] : 1.0;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = j0 + kk + 4 * reg;
            if (!GUARD || j < n) {
                double v = acc[reg] * rsv;
                if (g.cs.p) v *= g.cs.at(chain)[j];
                double* dst = C + ia + (long)n * j;
                if (g.accumulate) v += *dst;
                *dst = v;
            }
        }
    }
}

// ---- latency-optimised variant for the n <= ~600 sizes of this code ----
// One 256-thread block per 16x16 tile of C with K split over its 4 waves: 4 x more waves than the
// one-wave-per-tile kernel (at n = 256: 1024 waves = one per SIMD of the chip), every wave issues
// ALL of its operand loads (its quarter of K) before the first MFMA -- one L2 round trip instead of
// one per 16-deep K block -- and the four partial tiles meet in LDS.  K quarters are rounded up to a
// multiple of 16; out-of-range k contributes zeros.
template <bool TRANSA, int KQ /* k per wave, multiple of 16 */>
__global__ __launch_bounds__(256) void gemm_splitk_kernel(GemmDesc g, int tiles_per_dim) {
    __shared__ double part[4][4][64];                 // [wave][reg][lane]
    const int chain = blockIdx.y;
    const int n = g.n;
    const double* __restrict__ A = g.A.at(chain);
    const double* __restrict__ B = g.B.at(chain);
    double* __restrict__ C = g.C.at(chain);
    const double* __restrict__ ks = g.ks.p ? g.ks.at(chain) : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x % tiles_per_dim) * 16, j0 = (blockIdx.x / tiles_per_dim) * 16;
    const int r = lane & 15, kk = lane >> 4;
    const int ia = min(i0 + r, n - 1), jb = min(j0 + r, n - 1);      // clamped: loads stay unconditional, results masked
    const bool ia_ok = i0 + r < n, jb_ok = j0 + r < n;
    const int kbeg = wave * KQ;
    constexpr int NB = KQ / 16;
    double av[NB][4], bv[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = kbeg + 16 * b + 4 * kk + s;
            const int kc = min(k, n - 1);
            double a = TRANSA ? A[kc + (long)n * ia] : A[ia + (long)n * kc];
            double bb = B[kc + (long)n * jb];
            if (ks) a *= ks[kc];
            const bool ok = k < n;
            av[b][s] = (ok && ia_ok) ? a : 0.0;
            bv[b][s] = (ok && jb_ok) ? bb : 0.0;
        }
    }
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b][s], av[b][s], acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) part[wave][reg][lane] = acc[reg];
    __syncthreads();
    // wave w finishes accumulator register w: lane holds C[i0 + r][j0 + kk + 4*w]
    const double v4 = part[0][wave][lane] + part[1][wave][lane] + part[2][wave][lane] + part[3][wave][lane];
    const int j = j0 + kk + 4 * wave;
    if (ia_ok && j < n) {
        double v = v4;
        if (g.rs.p) v *= g.rs.at(chain)[i0 + r];
        if (g.cs.p) v *= g.cs.at(chain)[j];
        double* dst = C + (i0 + r) + (long)n * j;
        if (g.accumulate) v += *dst;
        *dst = v;
    }
}

template <int KQ>
static void launch_splitk(const GemmDesc& g, int n_chains, hipStream_t s) {
    const int tiles = (g.n + 15) / 16;
    dim3 grid(tiles * tiles, n_chains), block(256);
    if (g.transA) hipLaunchKernelGGL((gemm_splitk_kernel<true, KQ>), grid, block, 0, s, g, tiles);
    else hipLaunchKernelGGL((gemm_splitk_kernel<false, KQ>), grid, block, 0, s, g, tiles);
}

int launch_gemm(const GemmDesc& g, int n_chains, hipStream_t s) {
    const int n = g.n;
    static const bool force_v1 = getenv("DQMC_GEMM_V1") != nullptr;      // A/B switch
    if (!force_v1 && n <= 640) {
        const int kq = ((n + 3) / 4 + 15) / 16 * 16;                     // k per wave
        if (kq <= 16) launch_splitk<16>(g, n_chains, s);
        else if (kq <= 32) launch_splitk<32>(g, n_chains, s);
        else if (kq <= 48) launch_splitk<48>(g, n_chains, s);
        else if (kq <= 64) launch_splitk<64>(g, n_chains, s);
        else if (kq <= 96) launch_splitk<96>(g, n_chains, s);
        else if (kq <= 128) launch_splitk<128>(g, n_chains, s);
        else launch_splitk<160>(g, n_chains, s);
        DQ_HIP(hipGetLastError());
        return 0;
    }
    const int tiles = (n + 31) / 32;
    dim3 grid(tiles * tiles, n_chains), block(256);
    const bool guard = (n % 32) != 0;
    if (g.transA) {
        if (guard) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, s, g, tiles);
        else hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, 0, s, g, tiles);
    } else {
        if (guard) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, s, g, tiles);
        else hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, s, g, tiles);
    }
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
