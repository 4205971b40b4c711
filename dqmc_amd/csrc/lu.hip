// lu.hip -- the dense solves behind arma::solve / arma::log_det in
// stablelinalg::inv_I_plus_ldr and inv_I_plus_ldr_mul_ldr
// (source/stablelinalg.cpp:112,119-123,147,155), restated as LAPACK
// dgetrf (partial pivoting) + dgetrs.
//
//   (factorisation)  lu_blocked.hip: 32-column panels on chip, MFMA rank-32 trailing update.  Rows are NOT swapped:
//                    perm[k] records the pivot row of step k and the factors stay in the original row order.
//   lu_solve_kernel  solve A X = B for n right-hand sides.  Columns of X are
//                    independent: grid = (n/16 column tiles) x chains, each
//                    wave keeps 4 columns in registers (lane <-> pivot
//                    position) and runs the forward (unit L) and backward (U)
//                    substitutions with the pivot element broadcast by a lane
//                    shuffle: no LDS, no barriers.
#include "common.h"
#include "wave.h"
#include <cstdlib>

namespace dq {


// P A = L U with the factors in A's original row order: row perm[k] of LU holds
// U[k, k:] and L[k, :k].  Lane/register position r of x below is the PIVOT position.
// mode 0: B = X on entry (dense RHS, overwritten by the solution)
// mode 1: B = diag(dg) (X overwritten)
// mode 2: LU holds R = R1 * Pi^T of a to_LDR result (column perm[j] of R is column j of the
//         upper-triangular R1; source/stablelinalg.cpp:51-52); X = R^-1 diag(dg) = Pi R1^-1 diag(dg)
//         by back substitution only -- the permuted triangular solve that replaces the LU behind
//         arma::solve(X, R, diagmat) at source/stablelinalg.cpp:112,147.
#ifndef DQ_LU_SOLVE_PD
#define DQ_LU_SOLVE_PD 4      // factor columns in flight per wave: a step costs ~0.15 us of instructions, an L2 round trip ~1 us
#endif
// CW = right-hand-side columns per wave.  The substitution is a chain of n dependent steps whose cost is instructions per step: a
// single engine uses CW = 1 (4x the workgroups, the shortest step; measured -3 ms per cfg-3 sweep against CW = 4), engines with many
// chains keep CW = 4 (fewer factor-column fetches per right-hand side).
template <int NR, int CW>
__global__ __launch_bounds__(256) void lu_solve_kernel(CMat LUm, const int* perm_p, long perm_stride, Mat Xm, CVec dg, int mode, int n) {
    const int chain = blockIdx.y;
    const double* __restrict__ LU = LUm.at(chain);
    const int* __restrict__ perm = perm_p + (long)chain * perm_stride;
    double* __restrict__ X = Xm.at(chain);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 4 + wave) * CW;
    if (c0 >= n) return;
    double x[CW][NR]; int prow[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int r = lane + 64 * k;
        prow[k] = (r < n) ? perm[r] : 0;
#pragma unroll
        for (int cc = 0; cc < CW; ++cc) {
            const int c = c0 + cc;
            double v = 0.0;
            if (r < n && c < n) {
                if (mode == 0) v = X[prow[k] + (long)n * c];
                else if (mode == 1) v = (prow[k] == c) ? dg.at(chain)[c] : 0.0;
                else v = (r == c) ? dg.at(chain)[c] : 0.0;
            }
            x[cc][k] = v;
        }
    }
    const bool tri = mode == 2;
    const int kmax = tri ? min(c0 + CW - 1, n - 1) : n - 1;      // R1^-1 is upper triangular
    // Both substitutions are serial chains of n steps; per step the pivot element is broadcast with
    // v_readlane (k is wave-uniform; a ds_bpermute shuffle costs ~10x more) and the factor columns of
    // the next FOUR steps are already in flight (they do not depend on x): a one-deep prefetch left
    // every step waiting on L2.  The ring of four is unrolled by hand so its slots are static registers.
    constexpr int PD = DQ_LU_SOLVE_PD;
    // forward substitution, unit lower L
    if (!tri) {
        double lq[PD][NR];
        auto fetchL = [&](int slot, int k) {
            const long cb = (long)n * min(k, n - 1);
#pragma unroll
            for (int k2 = 0; k2 < NR; ++k2) { const int r = lane + 64 * k2; const double v = LU[prow[k2] + cb]; lq[slot][k2] = (r > k && r < n) ? v : 0.0; }
        };
#pragma unroll
        for (int e = 0; e < PD; ++e) fetchL(e, e);
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            for (int kb = 0; kb < 64; kb += PD) {
                if (64 * q + kb >= n) break;
#pragma unroll
                for (int e = 0; e < PD; ++e) {
                    const int k = 64 * q + kb + e;
                    double xk[CW];
#pragma unroll
                    for (int cc = 0; cc < CW; ++cc) xk[cc] = readlane_f64(x[cc][q], kb + e);
#pragma unroll
                    for (int k2 = q; k2 < NR; ++k2) {
#pragma unroll
                        for (int cc = 0; cc < CW; ++cc) x[cc][k2] -= lq[e][k2] * xk[cc];      // lq is 0 for rows <= k and for k >= n
                    }
                    fetchL(e, k + PD);
                }
            }
        }
    }
    // backward substitution, U
    {
        double uq[PD][NR]; double dq_[PD];
        auto fetchU = [&](int slot, int k) {
            const int kc = min(max(k, 0), n - 1);                           // clamped: out-of-range steps load a harmless valid column
            int sel = 0;
#pragma unroll
            for (int q2 = 0; q2 < NR; ++q2) if (q2 == (kc >> 6)) sel = prow[q2];
            const int pk = __builtin_amdgcn_readlane(sel, kc & 63);         // LU: pivot row of step k; tri: column of R holding R1[:, k]
            const long cb = tri ? (long)n * pk : (long)n * kc;
            dq_[slot] = tri ? LU[kc + cb] : LU[pk + cb];
#pragma unroll
            for (int k2 = 0; k2 < NR; ++k2) { const int r = lane + 64 * k2; const double v = LU[(tri ? min(r, n - 1) : prow[k2]) + cb]; uq[slot][k2] = (r < kc) ? v : 0.0; }
        };
        // steps run k = ktop .. 0 with ktop = 64*NR - 1 rounded so that slot e serves k == e' (mod 4) statically
        const int ktop = 64 * NR - 1;
#pragma unroll
        for (int e = PD - 1; e >= 0; --e) fetchU(e, ktop - (PD - 1 - e));
#pragma unroll
        for (int q = NR - 1; q >= 0; --q) {
            for (int kb = 64 - PD; kb >= 0; kb -= PD) {
#pragma unroll
                for (int e = PD - 1; e >= 0; --e) {
                    const int k = 64 * q + kb + e;
                    if (k <= kmax) {
                        const double rukk = 1.0 / dq_[e];
                        double xk[CW];
#pragma unroll
                        for (int cc = 0; cc < CW; ++cc) {
                            const double v = readlane_f64(x[cc][q], kb + e) * rukk;
                            xk[cc] = v;
                            if (lane == kb + e) x[cc][q] = v;
                        }
#pragma unroll
                        for (int k2 = 0; k2 <= q; ++k2) {
#pragma unroll
                            for (int cc = 0; cc < CW; ++cc) x[cc][k2] -= uq[e][k2] * xk[cc];    // uq is 0 for rows >= k
                        }
                    }
                    fetchU(e, k - PD);
                }
            }
        }
    }
#pragma unroll
    for (int cc = 0; cc < CW; ++cc) {
        if (c0 + cc >= n) continue;
#pragma unroll
        for (int k = 0; k < NR; ++k) { const int r = lane + 64 * k; if (r < n) X[(tri ? prow[k] : r) + (long)n * (c0 + cc)] = x[cc][k]; }
    }
}

int launch_lu_blocked(Mat A, int* perm, long perm_stride, int* rowpos, long rowpos_stride, double* logabsdet, int accumulate_logdet,
                      int* info, int n, int n_chains, hipStream_t s);     // lu_blocked.hip

int launch_lu(Mat A, int* perm, long perm_stride, double* logabsdet, int accumulate_logdet, int* info, int n, int n_chains, hipStream_t s,
              int* rowpos, long rowpos_stride) {
    if (n > 1024) { set_error("LU kernel supports n <= 1024"); return -1; }
    if (!rowpos) { set_error("LU: row-position workspace missing"); return -1; }
    return launch_lu_blocked(A, perm, perm_stride, rowpos, rowpos_stride, logabsdet, accumulate_logdet, info, n, n_chains, s);
}

template <int NR>
static int launch_solve_nr(CMat LU, const int* perm, long ps, Mat X, CVec dg, int mode, int n, int n_chains, hipStream_t s) {
    if (n_chains <= 4) hipLaunchKernelGGL((lu_solve_kernel<NR, 1>), dim3((n + 3) / 4, n_chains), dim3(256), 0, s, LU, perm, ps, X, dg, mode, n);
    else hipLaunchKernelGGL((lu_solve_kernel<NR, 4>), dim3((n + 15) / 16, n_chains), dim3(256), 0, s, LU, perm, ps, X, dg, mode, n);
    DQ_HIP(hipGetLastError());
    return 0;
}
int launch_lu_solve(CMat LU, const int* perm, long perm_stride, Mat X, CVec dg, int mode, int n, int n_chains, hipStream_t s) {
    if (n > 1024) { set_error("LU solve kernel supports n <= 1024"); return -1; }
    if (n <= 64) return launch_solve_nr<1>(LU, perm, perm_stride, X, dg, mode, n, n_chains, s);
    if (n <= 128) return launch_solve_nr<2>(LU, perm, perm_stride, X, dg, mode, n, n_chains, s);
    if (n <= 256) return launch_solve_nr<4>(LU, perm, perm_stride, X, dg, mode, n, n_chains, s);
    if (n <= 576) return launch_solve_nr<9>(LU, perm, perm_stride, X, dg, mode, n, n_chains, s);
    return launch_solve_nr<16>(LU, perm, perm_stride, X, dg, mode, n, n_chains, s);
}

}  // namespace dq
