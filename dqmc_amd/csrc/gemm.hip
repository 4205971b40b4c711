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
        if (g.CT.p) g.CT.at(chain)[j + (long)n * (i0 + r)] = v;       // transposed copy for the local-update walk (G -> GT)
    }
}

// ---- throughput variant for batched engines (many chains per launch) ----
// 64x64 tile of C per 256-thread block, 32x32 per wave (2x2 MFMA tiles), K walked in 16-deep stages
// through LDS (A stage [16][64] + B stage [16][64], rows padded to 80 doubles so the four k-groups
// of a wave read disjoint banks), next stage's global loads in flight during the MFMAs.
// Operand traffic per tile is 4x lower than one-wave-per-tile kernels: at 64 chains the split-K
// kernel is L2-bandwidth bound (13 TFLOP/s), this one is bound by the fp64 MFMA rate.
// Requires n % 64 == 0 (N = 64, 256, 576 of the reference configurations).
template <bool TRANSA>
__global__ __launch_bounds__(256) void gemm_tile64_kernel(GemmDesc g, int tiles_per_dim) {
    constexpr int LDT = 80;
    __shared__ double As[2][16][LDT];
    __shared__ double Bs[2][16][LDT];
    // XCD-aware tile mapping.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order and every XCD has its own L2:
    // with (tile, chain) = (blockIdx.x, blockIdx.y) the 16 tiles of one chain land on all 8 XCDs and each L2 pulls that chain's
    // A and B panels from the fabric separately -- 512 MB per 128-chain GEMM, which is what bounded the kernel at 105 us.
    // Here workgroup w serves logical slot (w % 8) * (total / 8) + w / 8, so a chain's tiles share one XCD and one L2.
    const int tiles2 = tiles_per_dim * tiles_per_dim;
    const int total = tiles2 * (int)gridDim.y;
    int lin = (int)blockIdx.x + tiles2 * (int)blockIdx.y;
    if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    const int chain = lin / tiles2, tile = lin % tiles2;
    const int n = g.n;
    const double* __restrict__ A = g.A.at(chain);
    const double* __restrict__ B = g.B.at(chain);
    double* __restrict__ C = g.C.at(chain);
    const double* __restrict__ ks = g.ks.p ? g.ks.at(chain) : nullptr;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i0 = (tile % tiles_per_dim) * 64, j0 = (tile / tiles_per_dim) * 64;
    const int r = lane & 15, kk = lane >> 4;
    const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;
    // staging roles
    const int ai = t & 63, akq = t >> 6;            // A: row ai, k = 4*akq + s
    const int bkp = t & 7, bj = t >> 3;             // B: k pair 2*bkp, columns bj and bj + 32
    double ra[4]; double2 rb0, rb1;
    auto load_stage = [&](int kb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = kb + 4 * akq + s;
            double v = TRANSA ? A[k + (long)n * (i0 + ai)] : A[(i0 + ai) + (long)n * k];
            if (ks) v *= ks[k];
            ra[s] = v;
        }
        rb0 = *reinterpret_cast<const double2*>(B + (kb + 2 * bkp) + (long)n * (j0 + bj));
        rb1 = *reinterpret_cast<const double2*>(B + (kb + 2 * bkp) + (long)n * (j0 + bj + 32));
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) As[buf][4 * akq + s][ai] = ra[s];
        Bs[buf][2 * bkp][bj] = rb0.x; Bs[buf][2 * bkp + 1][bj] = rb0.y;
        Bs[buf][2 * bkp][bj + 32] = rb1.x; Bs[buf][2 * bkp + 1][bj + 32] = rb1.y;
    };
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    load_stage(0); store_stage(0);
    __syncthreads();
    const int nstage = n / 16;
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstage) load_stage(16 * (st + 1));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + kk;
            const double a0 = As[buf][k][wi + r], a1 = As[buf][k][wi + 16 + r];
            const double b0 = Bs[buf][k][wj + r], b1 = Bs[buf][k][wj + 16 + r];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
        }
        if (st + 1 < nstage) store_stage(buf ^ 1);
        __syncthreads();
    }
    // lane holds C[i0 + wi + 16*ti + r][j0 + wj + 16*tj + kk + 4*reg]
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int i = i0 + wi + 16 * ti + r;
        const double rsv = g.rs.p ? g.rs.at(chain)[i] : 1.0;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = j0 + wj + 16 * tj + kk + 4 * reg;
                double v = acc[ti][tj][reg] * rsv;
                if (g.cs.p) v *= g.cs.at(chain)[j];
                double* dst = C + i + (long)n * j;
                if (g.accumulate) v += *dst;
                *dst = v;
            }
    }
}

template <int KQ>
static void launch_splitk(const GemmDesc& g, int n_chains, hipStream_t s) {
    const int tiles = (g.n + 15) / 16;
    dim3 grid(tiles * tiles, n_chains), block(256);
    if (g.transA) hipLaunchKernelGGL((gemm_splitk_kernel<true, KQ>), grid, block, 0, s, g, tiles);
    else hipLaunchKernelGGL((gemm_splitk_kernel<false, KQ>), grid, block, 0, s, g, tiles);
}

static int launch_gemm_main(const GemmDesc& g, int n_chains, hipStream_t s, bool* ct_done);

int launch_gemm(const GemmDesc& g, int n_chains, hipStream_t s) {
    bool ct_done = false;
    if (int rc = launch_gemm_main(g, n_chains, s, &ct_done)) return rc;
    if (g.CT.p && !ct_done) return launch_transpose_scale(CMat(g.C.p, g.C.stride), g.CT, CVec(), g.n, n_chains, s);
    return 0;
}

static int launch_gemm_main(const GemmDesc& g, int n_chains, hipStream_t s, bool* ct_done) {
    const int n = g.n;
    static const bool no_tile64 = getenv("DQMC_GEMM_NO_TILE64") != nullptr;
    if (!no_tile64 && n % 64 == 0 && (long)n_chains * (n / 64) * (n / 64) >= 256) {     // enough 64x64 tiles to fill 256 CUs
        const int tiles = n / 64;
        dim3 grid(tiles * tiles, n_chains), block(256);
        if (g.transA) hipLaunchKernelGGL((gemm_tile64_kernel<true>), grid, block, 0, s, g, tiles);
        else hipLaunchKernelGGL((gemm_tile64_kernel<false>), grid, block, 0, s, g, tiles);
        DQ_HIP(hipGetLastError());
        return 0;
    }
    if (n <= 640) {
        const int kq = ((n + 3) / 4 + 15) / 16 * 16;                     // k per wave
        if (kq <= 16) launch_splitk<16>(g, n_chains, s);
        else if (kq <= 32) launch_splitk<32>(g, n_chains, s);
        else if (kq <= 48) launch_splitk<48>(g, n_chains, s);
        else if (kq <= 64) launch_splitk<64>(g, n_chains, s);
        else if (kq <= 96) launch_splitk<96>(g, n_chains, s);
        else if (kq <= 128) launch_splitk<128>(g, n_chains, s);
        else launch_splitk<160>(g, n_chains, s);
        DQ_HIP(hipGetLastError());
        *ct_done = true;
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
