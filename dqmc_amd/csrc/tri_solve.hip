// tri_solve.hip -- X = R^-1 diag(dg) for the R of ONE to_LDR factorisation, blocked, on the matrix cores.
//
// The reference calls arma::solve(X, R, diagmat(..)) (source/stablelinalg.cpp:112,147).  R = R1 * Pi^T with R1 upper
// triangular (column jpvt[j] of R is column j of R1, source/stablelinalg.cpp:51-52), so X = Pi * (R1^-1 D) needs no
// factorisation, only a back substitution -- which lu_solve_kernel (mode 2) runs column by column: a chain of n dependent
// steps, 60 us at n = 256 and 385 us at n = 576 whatever the number of CUs.  Here the chain has n/16 steps:
//
//   Y = R1^-1 D is upper triangular.  In 16 x 16 blocks, block column c of Y is
//       for m = c .. 0:   Y_mc = T_m^-1 A_m,   A_b -= R1_bm Y_mc for all b < m          (A_c = D_c, A_b = 0 for b < c)
//   with T_m = R1_mm.  tri_inv_kernel inverts the diagonal blocks first (one column of an inverse per lane, 16 steps each,
//   all blocks at once).  tri_solve_kernel runs one workgroup per block column: wave w keeps the accumulators A_b, b = w
//   mod 4, in registers (MFMA D tiles), the owner of A_m forms Y_mc with four MFMAs and passes it on through LDS in the
//   layout in which a D tile is the next product's B operand (lane (r16, kk), register s <-> row kk + 4 s), one LDS-only
//   barrier per step, and every wave subtracts R1_bm Y_mc from its tiles, the tile of the next step first.  The R1 operands
//   of step m - 1 are loaded while step m is multiplied (they do not depend on Y).
//   Rows of X are scattered through jpvt exactly as the substitution kernel scatters them; rows below the diagonal block of
//   a column block are zero and are written as such.
#include "common.h"

namespace dq {

namespace {
using d4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ void lds_barrier_only() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
}  // namespace

// grid = (ceil(nb / 4), chains), 64 threads: 16 lanes per diagonal block, lane c of a group solves T x = e_c
__global__ __launch_bounds__(64) void tri_inv_kernel(CMat Rm, const int* perm_p, long perm_stride, double* tinv_p, long tinv_stride, int n) {
    __shared__ double T[4][16][17];
    const int chain = blockIdx.y;
    const double* __restrict__ R = Rm.at(chain);
    const int* __restrict__ perm = perm_p + (long)chain * perm_stride;
    double* tinv = tinv_p + (long)chain * tinv_stride;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int nb = (n + 15) / 16;
    const int m = 4 * blockIdx.x + g;
    const bool have = m < nb;
    // column c of the block: R1[16 m + j][16 m + c] = R[16 m + j + n * perm[16 m + c]]; identity beyond n
    const int col = 16 * m + c;
    const long cb = (have && col < n) ? (long)n * perm[col] : 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row = 16 * m + j;
        double v = (j == c) ? 1.0 : 0.0;
        if (have && col < n && row < n && j <= c) v = R[row + cb];
        T[g][j][c] = v;
    }
    __syncthreads();
    double x[16];
#pragma unroll
    for (int j = 15; j >= 0; --j) {
        double s0 = (j == c) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
        for (int i = j + 1; i < 16; ++i) {
            const double tt = T[g][j][i];
            if (i & 1) s1 = fma(-tt, x[i], s1); else s0 = fma(-tt, x[i], s0);
        }
        x[j] = (j <= c) ? (s0 + s1) / T[g][j][j] : 0.0;      // the inverse is upper triangular as well
    }
    if (have) {
        double* out = tinv + (long)m * 256 + 16 * c;          // column-major 16 x 16: Tinv[j + 16 c]
#pragma unroll
        for (int j = 0; j < 16; ++j) out[j] = x[j];
    }
}

// grid = (nb, chains), 256 threads.  NBW = ceil(nb / 4) accumulator tiles per wave.
template <int NBW>
__global__ __launch_bounds__(256) void tri_solve_kernel(CMat Rm, const int* perm_p, long perm_stride, Mat Xm, CVec dg, const double* tinv_p, long tinv_stride, int n) {
    __shared__ double Ybuf[2][4][64];
    __shared__ int pl[656];                                   // jpvt of the columns this block column touches (n <= 640)
    const int chain = blockIdx.y;
    const double* __restrict__ R = Rm.at(chain);
    const int* __restrict__ perm = perm_p + (long)chain * perm_stride;
    double* __restrict__ X = Xm.at(chain);
    const double* __restrict__ tinv = tinv_p + (long)chain * tinv_stride;
    const double* __restrict__ dgc = dg.at(chain);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int nb = (n + 15) / 16;
    const int c = nb - 1 - (int)blockIdx.x;                   // longest chains are dispatched first
    const int xcol = 16 * c + r16;                            // this lane's column of X (D layout: column r16)
    const bool col_ok = xcol < n;
    const long xoff = (long)n * (col_ok ? xcol : 0);
    // the pivot order goes to LDS once: read from memory inside the step loop it is a second dependent round trip per step
    // (jpvt -> address of the R1 operand), 2.3 us per step instead of one load latency
    for (int k = t; k < 16 * (c + 1); k += 256) pl[k] = perm[k < n ? k : n - 1];
    __syncthreads();
    // rows below the diagonal block of this block column are zero
    for (int e = t; e < 16 * (n - 16 * (c + 1)); e += 256) {
        const int k = 16 * (c + 1) + (e >> 4), cc = 16 * c + (e & 15);
        if (k < n && cc < n) X[perm[k] + (long)n * cc] = 0.0;
    }
    // accumulators: tile b = wave + 4 q  (q < NBW), D layout acc[reg] = A_b[row kk + 4 reg][col r16]
    d4 acc[NBW];
#pragma unroll
    for (int q = 0; q < NBW; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    {
        const double dv = col_ok ? dgc[xcol] : 1.0;
#pragma unroll
        for (int q = 0; q < NBW; ++q)
            if (wave + 4 * q == c) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[q][reg] = (kk + 4 * reg == r16) ? dv : 0.0;
            }
    }
    // operands of step m for this wave: ra[q][s] = R1[16 b + r16][16 m + kk + 4 s] for its tiles b = wave + 4 q < m
    double ra[2][NBW][4]; double ti[2][4]; int pr[2][4];
    auto fetch = [&](int m, double (&a)[NBW][4], double (&tv)[4], int (&prow)[4]) {
        const int mc = m < 0 ? 0 : m;
        long cbs[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * mc + kk + 4 * s;               // row of Y_m in D layout = column of R1 in the A operand
            const int pk = pl[k];
            prow[s] = pk; cbs[s] = (long)n * pk;
            tv[s] = tinv[(long)mc * 256 + r16 + 16 * (kk + 4 * s)];
        }
#pragma unroll
        for (int q = 0; q < NBW; ++q) {
            const int row = 16 * (wave + 4 * q) + r16;
            const int rc = row < n ? row : n - 1;
#pragma unroll
            // raw, unconditional, clamped loads: any arithmetic (or select) on the value here would make the wave wait for the
            // load it has just issued.  No mask is needed: tiles b >= m are never multiplied, rows of tiles b < m are < n, and a
            // column >= n (partial last block) meets a row of Y that is zero in every column that is stored
            for (int s = 0; s < 4; ++s) a[q][s] = R[rc + cbs[s]];
        }
    };
    fetch(c, ra[0], ti[0], pr[0]);
    // the step loop is unrolled by two so that the operand double buffer is indexed statically
    auto step = [&](int m, double (&a)[NBW][4], double (&tv)[4], int (&prow)[4], double (&an)[NBW][4], double (&tvn)[4], int (&prn)[4]) {
        fetch(m - 1, an, tvn, prn);                           // next step's operands: in flight during this step
        const int par = m & 1;
        if (wave == (m & 3)) {                                // Y_m = T_m^-1 A_m
            d4 y = {0.0, 0.0, 0.0, 0.0};
            d4 am = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NBW; ++q) if (q == (m >> 2)) am = acc[q];
#pragma unroll
            for (int s = 0; s < 4; ++s) y = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[s], am[s], y, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                Ybuf[par][reg][lane] = y[reg];
                if (col_ok && 16 * m + kk + 4 * reg < n) X[prow[reg] + xoff] = y[reg];
            }
        }
        lds_barrier_only();
        double yb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) yb[s] = Ybuf[par][s][lane];
        // A_b -= R1_bm Y_m, the tile of the next step (b = m - 1) first: tiles in descending order
#pragma unroll
        for (int q = NBW - 1; q >= 0; --q) {
            if (wave + 4 * q < m) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[q][s], yb[s], acc[q], 0, 0, 0);
            }
        }
    };
    int m = c;
    for (; m >= 1; m -= 2) {
        step(m, ra[0], ti[0], pr[0], ra[1], ti[1], pr[1]);
        step(m - 1, ra[1], ti[1], pr[1], ra[0], ti[0], pr[0]);
    }
    if (m == 0) step(0, ra[0], ti[0], pr[0], ra[1], ti[1], pr[1]);
}

// X = R^-1 diag(dg), R the permuted-triangular factor of one to_LDR (perm = its jpvt).  scratch: 16 * n doubles per chain.
int launch_tri_solve(CMat R, const int* perm, long perm_stride, Mat X, CVec dg, double* scratch, long scratch_stride, int n, int n_chains, hipStream_t s) {
    if (n > 640 || !scratch || scratch_stride < 16L * ((n + 15) / 16) * 16) { set_error("tri_solve: n <= 640 and 16 n doubles of scratch per chain"); return -1; }
    const int nb = (n + 15) / 16;
    hipLaunchKernelGGL(tri_inv_kernel, dim3((nb + 3) / 4, n_chains), dim3(64), 0, s, R, perm, perm_stride, scratch, scratch_stride, n);
    const dim3 grid(nb, n_chains), block(256);
    const int nbw = (nb + 3) / 4;
#define DQ_TS(K) hipLaunchKernelGGL((tri_solve_kernel<K>), grid, block, 0, s, R, perm, perm_stride, X, dg, (const double*)scratch, scratch_stride, n)
    if (nbw <= 1) DQ_TS(1); else if (nbw <= 2) DQ_TS(2); else if (nbw <= 4) DQ_TS(4); else if (nbw <= 6) DQ_TS(6); else if (nbw <= 9) DQ_TS(9); else DQ_TS(10);
#undef DQ_TS
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
