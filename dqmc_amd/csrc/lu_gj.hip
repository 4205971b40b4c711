// lu_gj.hip -- X = A^-1 B for n <= 256 (the arma::solve(M, RHS) of
// stablelinalg::inv_I_plus_ldr_mul_ldr / inv_invldr_plus_ldr / inv_I_plus_ldr,
// source/stablelinalg.cpp:122-125,153-155,184-186) as a blocked Gauss-Jordan
// elimination with partial pivoting, two launches per panel of 32 columns and
// no substitution phase at all:
//
//   gj_panel_kernel   ONE wave per chain.  Lane owns rows lane, lane+64, ... of the panel in
//                     registers and runs dgetf2 on the live rows (pivot search = per-lane max +
//                     DPP wave max, pivot row broadcast through LDS; a single wave needs no
//                     barrier).  A is NOT modified: the kernel emits the pivot rows S of the
//                     panel, log|det| of the pivot block P11 = L11 U11 and the two triangular
//                     inverses L11^-1, U11^-1 (32 x 32 each, one column per lane).
//   gj_update_kernel  whole chip, fp64 MFMA.  Every wave forms, for its 16 columns,
//                     U12 = U11^-1 (L11^-1 A[S, cols]) with the matrix cores (the pivot rows
//                     are read-only in this launch) and subtracts A[r, panel] U12 from every
//                     other row r -- the live rows of A and B, and the rows retired by earlier
//                     panels.  Retired rows live in pivot order in SA | X, so that after the
//                     last panel A has become the identity in that order and X holds the
//                     solution in natural row order: no row permutation, no triangular solves.
//
// The Schur complements are those of dgetrf with the same pivot choice inside a panel
// (A22 - A21 P11^-1 A12); against dgetrf + dgetrs the result differs at rounding level
// (measured on the cfg-3 stabilisation matrices, cond(M) ~ 1e3: 1e-13 .. 9e-13 on G, the same
// spread two LAPACK routes show among themselves).  For n = 256: 8 x (panel 13 us + update
// 8 us) instead of the 8 x (30 + 18 + 7) us of lu_blocked.hip plus a ~200 us substitution.
#include "common.h"
#include "wave.h"

namespace dq {

namespace {
constexpr int GJ_NB = 32;
using d4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ unsigned long long gj_key(double a, int r) {
    // |a| as an ordered unsigned integer; the low 10 bits carry 1023 - row so equal magnitudes pick the lowest row (idamax)
    return ((unsigned long long)__double_as_longlong(fabs(a)) & ~0x3FFULL) | (unsigned long long)(1023 - r) | (1ULL << 63);
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One elimination step of the panel, J a compile-time constant so that every register index is static
// (a rolled or partially unrolled step loop would put the panel into scratch).
template <int NR, int J>
__device__ __forceinline__ void gj_step(double (&a)[NR][GJ_NB], bool (&live)[NR], int (&mypos)[NR], double (*LU)[GJ_NB], int* prow_idx,
                                        int lane, int k0) {
    unsigned long long key = 0ULL;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const unsigned long long kq = live[q] ? gj_key(a[q][J], lane + 64 * q) : 0ULL;
        key = kq > key ? kq : key;
    }
    key = wave_max_u64(key);
    const int p = 1023 - (int)(key & 0x3FFULL);
    const int pl = p & 63, pq = p >> 6;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        if (q == pq && lane == pl) {             // the pivot row: L11[J][:J] and U11[J][J:] are final now
#pragma unroll
            for (int c = 0; c < GJ_NB; c += 2) *reinterpret_cast<double2*>(&LU[J][c]) = make_double2(a[q][c], a[q][c + 1]);
            live[q] = false; mypos[q] = k0 + J;
            prow_idx[J] = p;
        }
    }
    lds_wait();
    constexpr int C0 = J & ~1;
    double prow[GJ_NB - C0];
#pragma unroll
    for (int c = C0; c < GJ_NB; c += 2) { const double2 v = *reinterpret_cast<const double2*>(&LU[J][c]); prow[c - C0] = v.x; prow[c + 1 - C0] = v.y; }
    const double rpiv = 1.0 / prow[J - C0];      // dgetf2 scales by the reciprocal pivot as well
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        if (live[q]) {
            const double l = a[q][J] * rpiv;
            a[q][J] = l;
#pragma unroll
            for (int c = J + 1; c < GJ_NB; ++c) a[q][c] -= l * prow[c - C0];
        }
    }
}
template <int NR, int J>
struct GjSteps {
    static __device__ __forceinline__ void run(double (&a)[NR][GJ_NB], bool (&live)[NR], int (&mypos)[NR], double (*LU)[GJ_NB], int* prow_idx,
                                               int lane, int nbw, int k0) {
        if (J < nbw) gj_step<NR, J>(a, live, mypos, LU, prow_idx, lane, k0);      // wave-uniform
        GjSteps<NR, J + 1>::run(a, live, mypos, LU, prow_idx, lane, nbw, k0);
    }
};
template <int NR>
struct GjSteps<NR, GJ_NB> {
    static __device__ __forceinline__ void run(double (&)[NR][GJ_NB], bool (&)[NR], int (&)[NR], double (*)[GJ_NB], int*, int, int, int) {}
};
}  // namespace

// tinv: per chain 2 * 32 * 32 doubles, column-major: Linv[i + 32 c], then Uinv[i + 32 c]
template <int NR>
__global__ __launch_bounds__(64) void gj_panel_kernel(CMat Am, int* rowpos_p, long rowpos_stride, int* perm_p, long perm_stride, double* tinv_p,
                                                      double* logabsdet, int accumulate, int* info, int n, int k0) {
    __shared__ __attribute__((aligned(16))) double LU[GJ_NB][GJ_NB];      // pivot rows in pivot order: L11 below, U11 on and above the diagonal
    __shared__ int prow_idx[GJ_NB];
    const int chain = blockIdx.y;
    const double* __restrict__ A = Am.at(chain);
    int* rowpos = rowpos_p + (long)chain * rowpos_stride;
    int* perm = perm_p + (long)chain * perm_stride;
    double* tinv = tinv_p + (long)chain * 2 * GJ_NB * GJ_NB;
    const int lane = threadIdx.x;
    const int nbw = min(GJ_NB, n - k0);

    bool live[NR]; int mypos[NR];
    double a[NR][GJ_NB];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int r = lane + 64 * q;
        live[q] = r < n && (k0 == 0 ? true : rowpos[r] < 0);
        mypos[q] = -1;
#pragma unroll
        for (int c = 0; c < GJ_NB; ++c) a[q][c] = (live[q] && c < nbw) ? A[r + (long)n * (k0 + c)] : 0.0;
    }
    for (int e = lane; e < GJ_NB * GJ_NB; e += 64) (&LU[0][0])[e] = ((e >> 5) == (e & 31)) ? 1.0 : 0.0;   // identity padding for nbw < 32
    lds_wait();

    GjSteps<NR, 0>::run(a, live, mypos, LU, prow_idx, lane, nbw, k0);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int r = lane + 64 * q;
        if (r < n) { if (k0 == 0) rowpos[r] = mypos[q]; else if (mypos[q] >= 0) rowpos[r] = mypos[q]; }
    }
    lds_wait();
    if (lane < nbw) perm[k0 + lane] = prow_idx[lane];
    {                                                    // log|det P11| and singularity check
        const double pv = lane < nbw ? fabs(LU[lane][lane]) : 1.0;
        const double ls = wave_sum(log(pv));
        const bool bad = __any(!(pv > 0.0));
        if (lane == 0) {
            if (logabsdet) logabsdet[chain] = ((accumulate || k0 > 0) ? logabsdet[chain] : 0.0) + ls;
            if (info && bad) atomicOr(info, 1);
        }
    }
    // Triangular inverses, one column per lane: lanes 0..31 column c of L11^-1 (forward substitution), lanes 32..63
    // column c of U11^-1 run as the same forward recurrence on the index-reversed matrix.  Rows / columns >= nbw of
    // LU are the identity, so both inverses are identity-padded as well.
    {
        const bool up = lane >= 32;
        const int c = lane & 31;
        const int cc = up ? 31 - c : c;                  // unit vector position in recurrence order
        double x[GJ_NB];
#pragma unroll
        for (int j = 0; j < GJ_NB; ++j) {
            double s = (j == cc) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < j; ++m) {
                const double t = up ? LU[31 - j][31 - m] : LU[j][m];
                s -= t * x[m];
            }
            x[j] = up ? s / LU[31 - j][31 - j] : s;
        }
        double* out = tinv + (up ? GJ_NB * GJ_NB : 0) + 32 * c;
#pragma unroll
        for (int j = 0; j < GJ_NB; ++j) out[up ? 31 - j : j] = x[j];
    }
}

// MFMA operand convention (v_mfma_f64_16x16x4_f64, lane = (r16, kk) = (lane & 15, lane >> 4)):
//   mfma(a, b, acc): a = A[row r16][k kk], b = B[k kk][col r16], acc[reg] = D[row kk + 4 reg][col r16].
// A D tile therefore feeds the next product as its B operand directly: acc[reg] of lane (c, kk) is row k = kk + 4 reg
// of column c, and any partition of k into groups of four works as long as the A operand uses the same one.
//
// grid.x = col_tiles * row_tiles; 256 threads = 2 x 2 waves of 16 x 16.  Row tiles: [0, nA) rows of A / B (live rows
// only), [nA, nA + nS) retired rows (pivot order, SA / X), the last one writes the panel's own rows U12 into SA / X.
// Column tiles: [0, nCA) trailing columns of A, then n/32 tiles of B.
__global__ __launch_bounds__(256) void gj_update_kernel(Mat Am, Mat Bm, Mat SAm, Mat Xm, const int* rowpos_p, long rowpos_stride,
                                                        const int* perm_p, long perm_stride, const double* tinv_p, int n, int k0, int nA, int nS, int nCA) {
    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    double* __restrict__ B = Bm.at(chain);
    double* __restrict__ SA = SAm.at(chain);
    double* __restrict__ X = Xm.at(chain);
    const int* rowpos = rowpos_p + (long)chain * rowpos_stride;
    const int* perm = perm_p + (long)chain * perm_stride;
    const double* __restrict__ Linv = tinv_p + (long)chain * 2 * GJ_NB * GJ_NB;
    const double* __restrict__ Uinv = Linv + GJ_NB * GJ_NB;
    const int nbw = min(GJ_NB, n - k0);
    const int row_tiles = nA + nS + 1;
    const int rt = blockIdx.x % row_tiles, ct = blockIdx.x / row_tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int wr = wave & 1, wc = wave >> 1;
    const bool colA = ct < nCA;
    const int col = (colA ? k0 + nbw + 32 * ct : 32 * (ct - nCA)) + 16 * wc + r16;      // this lane's column (B operand / D layout)
    const bool col_ok = col < n;
    const long coff = (long)n * (col_ok ? col : n - 1);
    const double* __restrict__ src12 = colA ? A : B;         // pivot rows are read from here
    double* __restrict__ dstS = colA ? SA : X;               // retired rows of this column block

    // ---- operands, all loads issued up front ----
    double a12[2][4];                                        // A12[k = 16h + 4s + kk][col]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * h + 4 * s + kk;
            const int pr = perm[k0 + (k < nbw ? k : 0)];
            const double v = src12[pr + coff];
            a12[h][s] = (k < nbw && col_ok) ? v : 0.0;
        }
    // rows of this wave
    const bool writer = rt == nA + nS;
    const bool rowS = rt >= nA;
    const int row0 = (rowS ? 32 * (rt - nA) : 32 * rt) + 16 * wr;         // A rows: original index; S rows: pivot position
    const int rowlim = rowS ? k0 : n;
    const double* __restrict__ msrc = rowS ? SA : A;         // multipliers A[r, panel]
    double* __restrict__ dst = rowS ? dstS : (colA ? A : B);
    double mult[2][4];                                       // Mult[row0 + r16][k = 16h + kk + 4s]
    double cold[4];                                          // C[row0 + kk + 4 reg][col]
    bool st_ok[4];
    if (!writer) {
        const int mrow = min(row0 + r16, rowlim - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 16 * h + kk + 4 * s;
                const double v = msrc[mrow + (long)n * (k0 + (k < nbw ? k : 0))];
                mult[h][s] = k < nbw ? v : 0.0;
            }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = row0 + kk + 4 * reg;
            const int rc = min(r, rowlim - 1);
            st_ok[reg] = col_ok && r < rowlim && (rowS || rowpos[rc] < 0);
            cold[reg] = dst[rc + coff];
        }
    }
    // ---- T1 = L11^-1 A12 (lower triangular: the upper row half needs k < 16 only) ----
    d4 t1[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int hp = 0; hp < 2; ++hp)
#pragma unroll
        for (int h = 0; h <= hp; ++h)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double lv = Linv[(16 * hp + r16) + 32 * (16 * h + 4 * s + kk)];
                t1[hp] = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, a12[h][s], t1[hp], 0, 0, 0);
            }
    // ---- U12 = U11^-1 T1 (upper triangular: the lower row half needs k >= 16 only); B operand = t1[h][reg], k = 16h + kk + 4 reg ----
    d4 u[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int hp = 0; hp < 2; ++hp)
#pragma unroll
        for (int h = hp; h < 2; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double uv = Uinv[(16 * hp + r16) + 32 * (16 * h + kk + 4 * reg)];
                u[hp] = __builtin_amdgcn_mfma_f64_16x16x4f64(uv, t1[h][reg], u[hp], 0, 0, 0);
            }
    if (writer) {                                            // rows k0 + 16 wr + kk + 4 reg of SA / X receive U12
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = 16 * wr + kk + 4 * reg;
            if (i < nbw && col_ok) dstS[k0 + i + coff] = u[wr][reg];
        }
        return;
    }
    // ---- C -= Mult U12 ----
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(mult[h][s], u[h][s], acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
        if (st_ok[reg]) dst[row0 + kk + 4 * reg + coff] = cold[reg] - acc[reg];
}

// X = A^-1 B (n <= 256).  A and B are destroyed; SA: n*n scratch per chain; tinv: 2048 doubles per chain;
// perm / rowpos: n ints per chain.  logabsdet (optional) receives (+)= log|det A|; *info |= 1 on a zero / NaN pivot.
int launch_gj_solve(Mat A, Mat B, Mat X, Mat SA, double* tinv, int* perm, long perm_stride, int* rowpos, long rowpos_stride,
                    double* logabsdet, int accumulate_logdet, int* info, int n, int n_chains, hipStream_t s) {
    if (n > 256) { set_error("gj_solve supports n <= 256"); return -1; }
    for (int k0 = 0; k0 < n; k0 += GJ_NB) {
        const dim3 pg(1, n_chains);
        if (n <= 64) hipLaunchKernelGGL((gj_panel_kernel<1>), pg, dim3(64), 0, s, CMat(A), rowpos, rowpos_stride, perm, perm_stride, tinv, logabsdet, accumulate_logdet, info, n, k0);
        else if (n <= 128) hipLaunchKernelGGL((gj_panel_kernel<2>), pg, dim3(64), 0, s, CMat(A), rowpos, rowpos_stride, perm, perm_stride, tinv, logabsdet, accumulate_logdet, info, n, k0);
        else hipLaunchKernelGGL((gj_panel_kernel<4>), pg, dim3(64), 0, s, CMat(A), rowpos, rowpos_stride, perm, perm_stride, tinv, logabsdet, accumulate_logdet, info, n, k0);
        const int nbw = n - k0 < GJ_NB ? n - k0 : GJ_NB;
        const int nA = (n + 31) / 32, nS = k0 / 32;
        const int nCA = (n - k0 - nbw + 31) / 32, nCB = (n + 31) / 32;
        hipLaunchKernelGGL(gj_update_kernel, dim3((nA + nS + 1) * (nCA + nCB), n_chains), dim3(256), 0, s, A, B, SA, X, (const int*)rowpos, rowpos_stride,
                           (const int*)perm, perm_stride, (const double*)tinv, n, k0, nA, nS, nCA);
    }
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
