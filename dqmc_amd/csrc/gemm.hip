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

// ---- single-matrix variant for 256 < n <= 1024 (cfg 5: n = 576) ----
// The split-K kernel above keeps a quarter of K per wave in registers (160 VGPRs of operands at n = 576): two workgroups
// per CU, 1 296 workgroups = three rounds, each a full memory round trip with 80 eight-byte loads per lane -- 45 us for
// 0.38 GFLOP.  Here a workgroup owns a 32x32 tile of C (one 16x16 MFMA tile per wave) and walks K in 32-deep stages through
// LDS, so the two waves that share rows (columns) share the A (B) stage and the operand traffic halves; the global loads of the
// next three stages sit in a register queue while a stage is multiplied.  ceil(n/32)^2 workgroups, all resident at once.
// Measured at n = 576: 22.6 us (split-K 45.0).  What bounds it now is the operand volume through a CU's L1 (about 25 GB/s
// per CU with every miss slot busy, whatever the queue depth -- 2, 3 and 5 stages time the same, and so do 64-deep stages and
// two accumulator chains): 324 tiles on 256 CUs leave 68 CUs with two workgroups = 590 KB each.  A 48x32 tile (216 workgroups,
// 369 KB per CU) would bring it to ~14 us; not built.
// Workgroups are dealt to the 8 XCDs round-robin, so workgroup w takes tile (w % 8) * chunk + w / 8 of the column-major tile
// list: an XCD's L2 serves all of A but only ~1/8 of B.  C leaves in 128-byte row segments from the accumulators and (when
// asked for) C^T in 256-byte segments through LDS.
//   * The queue's loads are inline asm and its waits explicit (s_waitcnt vmcnt(2 x loads per stage)): left to the compiler the
//     wait in front of a stage's LDS write was vmcnt(0..7) on every path (the insertion pass merges the unrolled bodies'
//     counters conservatively), one full memory round trip per stage.  Every iteration issues the same number of loads (past
//     the end: a repeat of the last stage that is never used) so the count stays uniform, loaded values are touched only after
//     the wait, and the queue is drained before the epilogue may reuse its registers.
//   * The barriers order LDS only (s_waitcnt lgkmcnt(0); s_barrier): a __syncthreads() would wait for the whole queue.
//   * LDS layouts are chosen so that neither side conflicts: an operand that arrives with consecutive lanes along k (B, and A
//     when transposed) is kept [column][k] with a row pitch of 34 doubles (MFMA reads: slot 2 r + kk, a bijection per half
//     wave), A otherwise [k][row] with pitch 48.
//   * Addresses are base pointers per thread + stage * stride; only a partial last stage (n % 32 != 0) or an edge tile takes
//     the clamped / masked path.
__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <bool TRANSA, bool HAS_KS, int KS /* 32 or 64: depth of a stage */>
__global__ __launch_bounds__(256) void gemm_stage32_kernel(GemmDesc g, int T, int chunk) {
    constexpr int D = 3, NL = KS / 8, LPS = (HAS_KS ? 3 : 2) * NL, PK = KS + 2, PR = 48;
    __shared__ double As[TRANSA ? 32 * PK : KS * PR];
    __shared__ double Bs[32 * PK];
    __shared__ double Ct[32][33];
    const int chain = blockIdx.y;
    const int w = blockIdx.x;
    const int L = (w & 7) * chunk + (w >> 3);
    if (L >= T * T) return;                                  // workgroup-uniform: the grid is rounded up to a multiple of 8
    const int n = g.n;
    const int i0 = (L % T) * 32, j0 = (L / T) * 32;
    const double* __restrict__ A = g.A.at(chain);
    const double* __restrict__ B = g.B.at(chain);
    double* __restrict__ C = g.C.at(chain);
    const double* __restrict__ ks = HAS_KS ? g.ks.at(chain) : nullptr;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 15, kk = lane >> 4;
    const int wi = (wave & 1) * 16, wj = (wave >> 1) * 16;
    // staging roles.  A: row rlo, k = rhi + 8 s;  A^T and B: k = klo, row / column khi + (256 / KS) s
    const int rlo = t & 31, rhi = t >> 5;
    const int klo = t & (KS - 1), khi = t / KS;
    constexpr int XS = 256 / KS;
    const int nst = (n + KS - 1) / KS;
    const bool interior = i0 + 32 <= n && j0 + 32 <= n;
    const bool ragged = (n % KS) != 0;
    const double* pa[NL]; const double* pb[NL]; const double* pk[NL];
#pragma unroll
    for (int s = 0; s < NL; ++s) {
        pa[s] = TRANSA ? A + klo + (long)n * min(i0 + khi + XS * s, n - 1) : A + min(i0 + rlo, n - 1) + (long)n * (rhi + 8 * s);
        pb[s] = B + klo + (long)n * min(j0 + khi + XS * s, n - 1);
        pk[s] = HAS_KS ? (TRANSA ? ks + klo : ks + rhi + 8 * s) : nullptr;
    }
    const long sa = TRANSA ? KS : (long)KS * n;                // doubles per stage
    double qa[D][NL], qb[D][NL], qk[D][NL];
#define DQ_GLOAD(dst, ptr) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
    auto load_stage = [&](int st, double (&a)[NL], double (&b)[NL], double (&kf)[NL]) {
        if (ragged && st == nst - 1) {                        // partial last stage: clamp k (uniform branch, loads stay unconditional inside)
#pragma unroll
            for (int s = 0; s < NL; ++s) {
                const int ka = min(KS * st + (TRANSA ? klo : rhi + 8 * s), n - 1), kb = min(KS * st + klo, n - 1);
                const double* qpa = TRANSA ? A + ka + (long)n * min(i0 + khi + XS * s, n - 1) : A + min(i0 + rlo, n - 1) + (long)n * ka;
                const double* qpb = B + kb + (long)n * min(j0 + khi + XS * s, n - 1);
                DQ_GLOAD(a[s], qpa); DQ_GLOAD(b[s], qpb);
                if (HAS_KS) { const double* qpk = ks + ka; DQ_GLOAD(kf[s], qpk); }
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < NL; ++s) {
            const double* qpa = pa[s] + st * sa; const double* qpb = pb[s] + st * KS;
            DQ_GLOAD(a[s], qpa); DQ_GLOAD(b[s], qpb);
            if (HAS_KS) { const double* qpk = pk[s] + st * KS; DQ_GLOAD(kf[s], qpk); }
        }
    };
#undef DQ_GLOAD
    // ties the slot's registers to the wait: nothing may touch them before it
    auto wait_stage = [&](double (&a)[NL], double (&b)[NL], double (&kf)[NL]) {
#pragma unroll
        for (int s = 0; s < NL; ++s) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a[s]), "+v"(b[s]), "+v"(kf[s]) : "n"((D - 1) * LPS) : "memory");
    };
    static_assert((D - 1) * LPS <= 63, "vmcnt is a 6-bit counter");
    auto store_stage = [&](int st, const double (&a)[NL], const double (&b)[NL], const double (&kf)[NL]) {
        const bool masked = !interior || (ragged && st == nst - 1);       // uniform
#pragma unroll
        for (int s = 0; s < NL; ++s) {
            const int xk = rhi + 8 * s, xr = khi + XS * s;
            double va = a[s], vb = b[s];
            if (HAS_KS) va *= kf[s];
            if (masked) {
                const int k = KS * st + (TRANSA ? klo : xk), row = i0 + (TRANSA ? xr : rlo);
                if (!(k < n && row < n)) va = 0.0;
                if (!(KS * st + klo < n && j0 + xr < n)) vb = 0.0;
            }
            if (TRANSA) As[xr * PK + klo] = va; else As[xk * PR + rlo] = va;
            Bs[xr * PK + klo] = vb;
        }
    };
#pragma unroll
    for (int u = 0; u < D; ++u) load_stage(min(u, nst - 1), qa[u], qb[u], qk[u]);
    d4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};      // two chains: a dependent fp64 MFMA waits for its predecessor's full latency
    for (int st0 = 0; st0 < nst; st0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int st = st0 + u;
            if (st < nst) {                                   // workgroup-uniform
                wait_stage(qa[u], qb[u], qk[u]);
                lds_only_barrier();                           // the previous stage has been read by every wave
                store_stage(st, qa[u], qb[u], qk[u]);
                load_stage(min(st + D, nst - 1), qa[u], qb[u], qk[u]);
                lds_only_barrier();
#pragma unroll
                for (int q = 0; q < KS / 4; ++q) {
                    const int k = 4 * q + kk;
                    const double av = TRANSA ? As[(wi + r) * PK + k] : As[k * PR + wi + r];
                    if (q & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Bs[(wj + r) * PK + k], av, acc2, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Bs[(wj + r) * PK + k], av, acc, 0, 0, 0);
                }
            }
        }
    }
    // drain the queue: its registers may be reused from here on
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
        for (int s = 0; s < NL; ++s) asm volatile("s_waitcnt vmcnt(0)" : "+v"(qa[u][s]), "+v"(qb[u][s]), "+v"(qk[u][s]) :: "memory");
    // lane holds C[i0 + wi + r][j0 + wj + kk + 4 reg]
    const int i = i0 + wi + r;
    const double rsv = (g.rs.p && i < n) ? g.rs.at(chain)[i] : 1.0;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int jl = wj + kk + 4 * reg, j = j0 + jl;
        double v = (acc[reg] + acc2[reg]) * rsv;
        if (i < n && j < n) {
            if (g.cs.p) v *= g.cs.at(chain)[j];
            double* dst = C + i + (long)n * j;
            if (g.accumulate) v += *dst;
            *dst = v;
        }
        Ct[wi + r][jl] = v;
    }
    if (g.CT.p) {
        __syncthreads();
        double* __restrict__ CT = g.CT.at(chain);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int il = rhi + 8 * s;                       // row of C = column of CT; rlo runs along j
            if (i0 + il < n && j0 + rlo < n) CT[(j0 + rlo) + (long)n * (i0 + il)] = Ct[il][rlo];
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
    if (n % 64 == 0 && (long)n_chains * (n / 64) * (n / 64) >= 256) {     // enough 64x64 tiles to fill 256 CUs
        const int tiles = n / 64;
        dim3 grid(tiles * tiles, n_chains), block(256);
        if (g.transA) hipLaunchKernelGGL((gemm_tile64_kernel<true>), grid, block, 0, s, g, tiles);
        else hipLaunchKernelGGL((gemm_tile64_kernel<false>), grid, block, 0, s, g, tiles);
        DQ_HIP(hipGetLastError());
        return 0;
    }
    if (n >= 384) {                                                 // measured: 13.7 vs 10.9 us at n = 324, 15 vs 22 at 400, 31 vs 45 at 576
        const int T = (n + 31) / 32, chunk = (T * T + 7) / 8;
        dim3 grid(8 * chunk, n_chains), block(256);
        // (64-deep stages measured no faster: 24.9 against 23.0 us at n = 576, with 250 VGPRs)
#define DQ_ST(TA, KSC) hipLaunchKernelGGL((gemm_stage32_kernel<TA, KSC, 32>), grid, block, 0, s, g, T, chunk)
        if (g.transA) { if (g.ks.p) DQ_ST(true, true); else DQ_ST(true, false); } else { if (g.ks.p) DQ_ST(false, true); else DQ_ST(false, false); }
#undef DQ_ST
        DQ_HIP(hipGetLastError());
        *ct_done = true;
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
