// lu_blocked.hip -- blocked right-looking LU with partial pivoting (dgetrf, the
// factorisation behind arma::solve / arma::log_det at source/stablelinalg.cpp:119-123,
// 155) in three kernels per panel of NB = 32 columns:
//
//   lu_panel_kernel   one 1024-thread workgroup per chain; thread r owns row r of the
//                     panel in REGISTERS (32 doubles), so the NB elimination steps of a
//                     panel touch neither HBM nor L2: the pivot search is a DPP
//                     wave-max + one LDS hop, the pivot row is broadcast through
//                     LDS, the step loop is fully unrolled (static register indices).
//   lu_trsm_kernel    U12 = L11^-1 A12 on the NB pivot rows, one thread per trailing
//                     column.
//   lu_update_kernel  A22 -= L21 U12 as a rank-NB update on the fp64 matrix cores,
//                     whole chip (v_mfma_f64_16x16x4, 16x16 tile per wave).
//
// As in lu.hip rows are never swapped: perm[k] is the pivot row of step k and the
// factors stay in A's original row order (row perm[k] holds U[k,k:] and L[k,:k]);
// lu_solve_kernel reads them through perm.  The single-workgroup streaming
// lu_kernel it replaces was bound by one CU's L1 bandwidth (2.6 ms at n = 256).
#include "common.h"
#include "wave.h"

namespace dq {

namespace {
constexpr int LU_NB = 32;

__device__ __forceinline__ unsigned long long piv_key(double a, int r) {
    // |a| ordered as an unsigned integer; low 10 bits carry 1023 - row so equal magnitudes pick the lowest row (idamax)
    return ((unsigned long long)__double_as_longlong(fabs(a)) & ~0x3FFULL) | (unsigned long long)(1023 - r);
}
// workgroup barrier that orders LDS traffic only (no vmcnt(0): see update.hip)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
}  // namespace

// rowpos: [chains][n] ints, -1 = live row, else the elimination step that made it a pivot row.
__global__ __launch_bounds__(1024) void lu_panel_kernel(Mat Am, int* perm_p, long perm_stride, int* rowpos_p, long rowpos_stride,
                                                        double* logabsdet, int accumulate, int* info, int n, int k0) {
    __shared__ __attribute__((aligned(16))) double prow[LU_NB];
    __shared__ double pivs[LU_NB];
    __shared__ unsigned long long keys[16];
    __shared__ int perml[LU_NB];
    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    int* perm = perm_p + (long)chain * perm_stride;
    int* rowpos = rowpos_p + (long)chain * rowpos_stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nbw = min(LU_NB, n - k0);
    const int nwaves = blockDim.x >> 6;
    const bool inb = t < n;
    bool live = inb && (k0 == 0 ? true : rowpos[t] < 0);
    double a[LU_NB];
#pragma unroll
    for (int c = 0; c < LU_NB; ++c) a[c] = (inb && c < nbw) ? A[t + (long)n * (k0 + c)] : 0.0;
    int my_pos = -1;

#pragma unroll
    for (int j = 0; j < LU_NB; ++j) {
        if (j < nbw) {                                   // uniform
            unsigned long long key = live ? piv_key(a[j], t) | (1ULL << 63) : 0ULL;
            key = wave_max_u64(key);
            if (lane == 0) keys[wave] = key;
            lds_barrier();
            unsigned long long best = keys[0];
            for (int q = 1; q < nwaves; ++q) { const unsigned long long o = keys[q]; best = o > best ? o : best; }
            const int p = 1023 - (int)(best & 0x3FFULL);
            if (t == p) {                                // publish the part of the pivot row still needed
                // an LDS store costs ~14-28 clk whatever the number of active lanes: publish in 16-byte pairs (even start)
#pragma unroll
                for (int c = (j & ~1); c < LU_NB; c += 2) *reinterpret_cast<double2*>(prow + c) = make_double2(a[c], a[c + 1]);
                pivs[j] = a[j];
                live = false; my_pos = k0 + j;
                perml[j] = p;                            // to memory after the loop: a global store here put a vmcnt(0) wait into the next barrier
            }
            lds_barrier();
            const double rpiv = 1.0 / prow[j];           // dgetf2 scales by the reciprocal pivot as well
            if (live) {
                const double l = a[j] * rpiv;
                a[j] = l;
#pragma unroll
                for (int c = j + 1; c < LU_NB; ++c) a[c] -= l * prow[c];
            }
        }
    }
    if (inb) {
#pragma unroll
        for (int c = 0; c < LU_NB; ++c) if (c < nbw) A[t + (long)n * (k0 + c)] = a[c];
        if (k0 == 0) rowpos[t] = my_pos; else if (my_pos >= 0) rowpos[t] = my_pos;
    }
    if (t < nbw) perm[k0 + t] = perml[t];
    __syncthreads();
    if (wave == 0) {                                     // log|det| contribution and singularity check, off the critical path
        const double pv = lane < nbw ? fabs(pivs[lane]) : 1.0;
        const double ls = wave_sum(log(pv));
        const bool bad = __any(!(pv > 0.0));
        if (lane == 0) {
            if (logabsdet) logabsdet[chain] = ((accumulate || k0 > 0) ? logabsdet[chain] : 0.0) + ls;
            if (info && bad) atomicOr(info, 1);
        }
    }
}

// U12[:, c] = L11^-1 A12[:, c] on the panel's pivot rows; thread <-> trailing column c.  The column's
// NB entries live in LDS (u[j][thread], conflict-free) so the substitution loops stay rolled.
__global__ __launch_bounds__(256) void lu_trsm_kernel(Mat Am, const int* perm_p, long perm_stride, int n, int k0) {
    __shared__ double L11[LU_NB][LU_NB + 1];
    __shared__ double u[LU_NB][256];
    __shared__ int prow[LU_NB];
    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    const int* perm = perm_p + (long)chain * perm_stride;
    const int nbw = min(LU_NB, n - k0);
    const int tid = threadIdx.x;
    for (int e = tid; e < LU_NB * LU_NB; e += blockDim.x) {
        const int j = e % LU_NB, m = e / LU_NB;          // L11[j][m] = A[perm[k0+j], k0+m]
        L11[j][m] = (j < nbw && m < j) ? A[perm[k0 + j] + (long)n * (k0 + m)] : 0.0;
    }
    if (tid < LU_NB) prow[tid] = tid < nbw ? perm[k0 + tid] : 0;
    __syncthreads();
    const int c = k0 + nbw + blockIdx.x * blockDim.x + tid;
    if (c >= n) return;
    for (int j = 0; j < nbw; ++j) u[j][tid] = A[prow[j] + (long)n * c];
    for (int j = 1; j < nbw; ++j) {
        double s = u[j][tid];
        for (int m = 0; m < j; ++m) s -= L11[j][m] * u[m][tid];
        u[j][tid] = s;
        A[prow[j] + (long)n * c] = s;
    }
}

// A22[r, c] -= sum_j L21[r, j] U12[j, c] for live rows r and columns c >= k0 + nbw.
__global__ __launch_bounds__(256) void lu_update_kernel(Mat Am, const int* perm_p, long perm_stride, const int* rowpos_p, long rowpos_stride,
                                                        int n, int k0, int row_tiles) {
    using d4 = __attribute__((ext_vector_type(4))) double;
    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    const int* perm = perm_p + (long)chain * perm_stride;
    const int* rowpos = rowpos_p + (long)chain * rowpos_stride;
    const int nbw = min(LU_NB, n - k0);
    const int c_first = k0 + nbw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x % row_tiles) * 32 + (wave & 1) * 16;
    const int j0 = c_first + (blockIdx.x / row_tiles) * 32 + (wave >> 1) * 16;
    if (i0 >= n || j0 >= n) return;
    const int r16 = lane & 15, kk = lane >> 4;
    const int row = i0 + r16, col = j0 + r16;
    const bool row_live = row < n && rowpos[row] < 0;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < LU_NB / 4; ++s) {
        const int k = 4 * s + kk;
        const double uv = (k < nbw && col < n) ? A[perm[k0 + k] + (long)n * col] : 0.0;      // U12[k][col]
        const double lv = (k < nbw && row_live) ? A[row + (long)n * (k0 + k)] : 0.0;         // L21[row][k]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(uv, lv, acc, 0, 0, 0);
    }
    // lane holds C[i0 + r16][j0 + kk + 4*reg]
    if (row_live) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = j0 + kk + 4 * reg;
            if (c < n) A[row + (long)n * c] -= acc[reg];
        }
    }
}

int launch_lu_blocked(Mat A, int* perm, long perm_stride, int* rowpos, long rowpos_stride, double* logabsdet, int accumulate_logdet,
                      int* info, int n, int n_chains, hipStream_t s) {
    if (n > 1024) { set_error("LU kernel supports n <= 1024"); return -1; }
    for (int k0 = 0; k0 < n; k0 += LU_NB) {
        // one thread per row; the per-wave serial work (DPP key-max, pivot-row broadcast) costs VALU time once per
        // resident wave, so fewer waves = a shorter step (microbench: 0.44 us at 4 waves vs 0.80 us at 16)
        hipLaunchKernelGGL(lu_panel_kernel, dim3(1, n_chains), dim3(((n + 63) / 64) * 64), 0, s, A, perm, perm_stride, rowpos, rowpos_stride,
                           logabsdet, accumulate_logdet, info, n, k0);
        const int nbw = n - k0 < LU_NB ? n - k0 : LU_NB;
        const int ntrail = n - k0 - nbw;
        if (ntrail > 0) {
            hipLaunchKernelGGL(lu_trsm_kernel, dim3((ntrail + 255) / 256, n_chains), dim3(256), 0, s, A, (const int*)perm, perm_stride, n, k0);
            const int row_tiles = (n + 31) / 32, col_tiles = (ntrail + 31) / 32;
            hipLaunchKernelGGL(lu_update_kernel, dim3(row_tiles * col_tiles, n_chains), dim3(256), 0, s, A, (const int*)perm, perm_stride,
                               (const int*)rowpos, rowpos_stride, n, k0, row_tiles);
        }
    }
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
