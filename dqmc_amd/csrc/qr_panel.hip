// qr_panel.hip -- stablelinalg::to_LDR (source/stablelinalg.cpp:35-55) as a PANEL-pivoted blocked Householder QR.
//
// The reference factorises with dgeqp3 (arma::qr(Q, R, P, M, "vector")): one global pivot decision per COLUMN, i.e. n dependent
// chip-wide steps (qr_colown.hip runs them on one CU in 3.2 us each, qr_coop.hip pays a cross-CU exchange for each).  to_LDR's
// consumers only ever use the product L diag(d) R and the grading of d, so the pivot ORDER is free as long
// as the factorisation stays rank-revealing.  Here one global decision is taken per panel of QP_B = 16 columns:
//
//   sketch     Y = Omega . A_trailing, Omega = QP_SR x m matrix of +-1 from an integer hash of the row index (fixed: the
//              factorisation is deterministic), formed on the matrix cores by the update kernel from the tiles it has just
//              updated -- the sketch is FRESH for every panel (a down-dated sketch loses the small columns of a graded DQMC
//              matrix to cancellation);
//   selection  the first 16 pivots of a column-pivoted Householder QR of the 32 x n_c sketch (one workgroup, lane <-> column, the
//              column in registers, one barrier per step) are the panel's columns, in that order;
//   panel      unpivoted Householder QR of the 16 selected full-height columns: lane <-> (column c = lane & 15, 16 rows), the
//              pivot column reaches the other 15 lanes of its DPP row by row_newbcast (v_fmac_f64_dpp), so the 15 dot products
//              and the 15 updates of a step are 16 DPP-FMAs each; ONE fused reduction per step yields the dots, the squared
//              norm of the pivot column's tail and the entries v_c^T v_j of the compact-WY factor T;
//   update     A <- (I - V T V^T)^T A on the trailing columns, 16 columns per workgroup, v_mfma_f64_16x16x4_f64;
//   form Q     Q = H_0 ... H_{n/16-1} from the same compact-WY factors (every panel's clean reflectors and T are kept), MFMA.
//
// n / 16 panels x (qp_panel_kernel, qp_update_kernel) + qp_formq_kernel: 575 us at n = 256 (qr_colown + formq_blocked: 892), 2.0 ms at
// n = 576 (qr_coop: 4.26).  Where the time goes, and the forms that were measured and dropped (one persistent launch, a lane pair per
// sketch column, 24 sketch rows, look-ahead selection): DESIGN.md section 5.
//
// Randomised panel pivoting (Duersch & Gu 2017; Martinsson, Quintana-Orti, Heavner, van de Geijn 2017).  The numpy statement
// of exactly this algorithm is oracle/panel_qr.py::qr_sketch(b = 16, p = 16, sign = True, local_pivot = False); its effect on
// G (cfg 3 thermalised sweep 3e-11 absolute, cfg 3 / cfg 5 i.i.d. <= 3e-11 relative; tournament pivoting and Gaussian sketches
// beside it) is in profiles/r04_eval_panel_qr_numpy.log; tests/test_panel_qr.py pins the numpy statements on the CPU.
// Output format = the other QRCP kernels': reflectors and R0 in place in A WITHOUT column swaps, tau, jpvt (formq_* and
// assemble_r_kernel of qr.hip finish L, d, R).
#include "common.h"
#include "wave.h"
#include <utility>

namespace dq {

namespace {

#ifdef DQ_QP_STAMPS
#define QST(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (threadIdx.x == 0) qst[i] = t_; }
#else
#define QST(i)
#endif

constexpr int QP_B = 16;         // panel width
constexpr int QP_SR = 32;        // sketch rows formed by the update kernel (two MFMA row tiles)
#ifndef QP_SEL_ROWS
#define QP_SEL_ROWS 32
#endif
constexpr int QP_SEL = QP_SEL_ROWS;   // ... and used by the selection (b + p, p = 16).  24 rows (p = 8) are 3 us per panel faster at n = 256 (25.2 against 28.2 us) but put the thermalised unequal-time series of cfg 3 at 1.44e-10 of its largest entry (32 rows: 6.8e-11; bound 1e-10)
typedef double d4 __attribute__((ext_vector_type(4)));

// Omega[i][r] = +-1: bit i of a 32-bit mix of the ROW index r (oracle/panel_qr.py::omega_sign is the same function)
__device__ __forceinline__ unsigned qp_row_bits(unsigned r) {
    unsigned h = r * 0x9E3779B1u + 0x85EBCA77u;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ double qp_sign(unsigned bits, int i) {
    return __hiloint2double((int)(0x3FF00000u | (((bits >> i) & 1u) << 31)), 0);
}

// lane-wise sum over the four 16-lane rows of a wave (gfx950 v_permlane16_swap / v_permlane32_swap), result in every lane
__device__ __forceinline__ double rows4_sum(double x) {
    const unsigned lo = (unsigned)__double_as_longlong(x), hi = (unsigned)(__double_as_longlong(x) >> 32);
    auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double e = __hiloint2double((int)h16[0], (int)l16[0]), o = __hiloint2double((int)h16[1], (int)l16[1]);   // even row | odd row of each pair
    const double s = e + o;
    const unsigned slo = (unsigned)__double_as_longlong(s), shi = (unsigned)(__double_as_longlong(s) >> 32);
    auto l32 = __builtin_amdgcn_permlane32_swap(slo, slo, false, false);
    auto h32 = __builtin_amdgcn_permlane32_swap(shi, shi, false, false);
    return __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);               // lower half | upper half
}

// d += a[lane J of this 16-lane row] * a  (own value times the broadcast one, one register).  No wait state in front: the callers
// fence every register these blocks read (dpp_fence) after its last VALU write.
template <int J>
__device__ __forceinline__ void dpp_fmac_self(double& d, double a) {
    asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(a), "n"(J));
}
// a += a[lane J] * m
template <int J>
__device__ __forceinline__ void dpp_axpy_self(double& a, double m) {
    asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m), "n"(J));
}
// every VALU write of a[] issued so far is complete and two wait states old: what a DPP read of these registers needs (the hazard
// recogniser does not look inside inline asm)
__device__ __forceinline__ void dpp_fence(double (&a)[16]) {
    asm volatile("s_nop 1" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                             "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}

__device__ __forceinline__ void dpp_fence1(double& z) { asm volatile("s_nop 1" : "+v"(z)); }
// d += z[lane T of this 16-lane row] * y
template <int T>
__device__ __forceinline__ void dpp_fmac(double& d, double z, double y) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(z), "v"(y), "n"(T));
}
// tj += sum over t < J of trow[t] * z[lane t of this row]
template <int... Ts>
__device__ __forceinline__ void t_row_dot(double& tj, const double (&trow)[16], double z, std::integer_sequence<int, Ts...>) {
    (dpp_fmac<Ts>(tj, z, trow[Ts]), ...);
}

template <int NW>
struct PanelShared {
    static constexpr int NWP = NW < 2 ? 2 : NW;
    alignas(16) unsigned long long key[2][NWP];   // selection: every wave's best {norm^2 | column}, double-buffered on the step parity
    alignas(16) double cand[2][QP_SEL + 2];        // ... and the winning column's residual sketch (rows j..31) | its tail norm^2
    alignas(16) double part[2][QP_B][NWP];        // panel: per wave partial x^T a_c, [column][wave]
    alignas(16) double prow[2][QP_B];             // panel: row k + j of the panel before step j
    int sel[QP_B];
};

// dlarfg for (alpha, sum of squares below): H = I - tau v v^T, v = (1, x * scale), H (alpha, x) = (beta, 0).
// 1 / sqrt and 1 / (alpha - beta) by v_rsq_f64 / v_rcp_f64 + NEWTON Newton steps instead of the IEEE sqrt and two divisions (60
// dependent instructions on every lane, on the critical path of every step): NEWTON = 2 is good to an ulp or two -- tau, beta and
// the reflector stay consistent to working precision, which is what the orthogonality of H needs --, NEWTON = 1 (sketch: the
// selection only ranks columns) to ~1e-8.
template <int NEWTON>
__device__ __forceinline__ void householder(double alpha, double tail2, double& beta, double& tau, double& scale) {
    const double s = fma(alpha, alpha, tail2);
    double r = __builtin_amdgcn_rsq(s);
    const double h = 0.5 * s;
#pragma unroll
    for (int it = 0; it < NEWTON; ++it) r = r * fma(-h * r, r, 1.5);
    double sq = s * r;
    if (NEWTON > 1) sq = fma(0.5 * r, fma(-sq, sq, s), sq);
    const double b = -copysign(sq, alpha);
    const double d = alpha - b;                                        // |alpha| + sqrt(s): no cancellation
    double id = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int it = 0; it < NEWTON; ++it) id = id * fma(-d, id, 2.0);
    const bool none = tail2 == 0.0;                                    // H = I
    beta = none ? alpha : b;
    tau = none ? 0.0 : d * copysign(r, alpha);                         // (beta - alpha) / beta = -d / beta = d * sign(alpha) / sqrt(s)
    scale = none ? 0.0 : id;
}

// the steps exchange data through LDS only: a barrier that does not drain the vector memory queue (__syncthreads() is s_waitcnt vmcnt(0)
// lgkmcnt(0) + s_barrier: it would park every wave until its global stores / loads have completed)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- selection: step J of the column-pivoted QR of the sketch ----
// A lane owns CPL columns, col_q = t + 64 NW q (q < CPL), each as QP_SEL registers: the workgroup stays at one wave per SIMD up to
// n = 1024 (NW = 4, CPL = 4).  With one column per lane n = 576 needs 9-10 waves, three on a SIMD, and a step costs what the three
// issue one after the other (24-row selection: 44.3 us per panel against 38.3 for 3 waves x 3 columns).
template <int J, int NW, int CPL>
__device__ __forceinline__ void select_step(double (&y)[CPL][QP_SEL], unsigned& live, int t, int lane, int wave, PanelShared<NW>& sh) {
    constexpr int par = J & 1;
    unsigned long long key = 0ULL; int bq = 0; double btail = 0.0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        double tl[4] = {0.0, 0.0, 0.0, 0.0};                         // four chains: a dependent fp64 FMA every ~8 clk, an independent one every 4
#pragma unroll
        for (int i = J + 1; i < QP_SEL; ++i) tl[i & 3] = fma(y[q][i], y[q][i], tl[i & 3]);
        const double tail = (tl[0] + tl[1]) + (tl[2] + tl[3]);
        const double nrm2 = fma(y[q][J], y[q][J], tail);
        // {norm^2 with its low 11 bits dropped | 1024 - column}: one maximum decides, the lowest column wins a tie
        const unsigned long long kq = ((live >> q) & 1u) ? (((unsigned long long)__double_as_longlong(nrm2) & ~0x7FFULL) | (unsigned long long)(1024 - (t + 64 * NW * q))) : 0ULL;
        if (kq > key) { key = kq; bq = q; btail = tail; }
    }
    const unsigned long long wmax = wave_max_u64(key);
    // Two barriers per step: every wave's candidate publishes its KEY, all waves pick the winner, and only the winning wave's candidate lane
    // publishes its column.  Publishing every candidate's column before the winner is known saved the second barrier but put NW columns on the
    // CU's one LDS pipeline per step (an LDS store costs the pipeline the same whether one lane is active or 64):
    // 26.7 -> 25.2 us per panel at N = 256, 40.3 -> 39.0 at N = 576.
    const bool cand_ = wmax != 0ULL && key == wmax;
    if (cand_) sh.key[par][wave] = wmax;
    if (wmax == 0ULL && lane == 0) sh.key[par][wave] = 0ULL;
    lds_barrier();
    unsigned long long best = sh.key[par][0]; int ww = 0;
#pragma unroll
    for (int q = 1; q < NW; ++q) { const unsigned long long o = sh.key[par][q]; if (o > best) { best = o; ww = q; } }
    if (cand_ && wave == ww) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            if (bq == q) {
#pragma unroll
                for (int i = J & ~1; i < QP_SEL; i += 2) *reinterpret_cast<double2*>(&sh.cand[par][i]) = double2{y[q][i], y[q][i + 1]};
            }
        }
        sh.cand[par][QP_SEL] = btail;
    }
    lds_barrier();
    const int pcol = 1024 - (int)(best & 0x7FFULL);
    const double* xs = sh.cand[par];
    double x[QP_SEL];
#pragma unroll
    for (int i = J & ~1; i < QP_SEL; i += 2) { const double2 v = *reinterpret_cast<const double2*>(&xs[i]); x[i] = v.x; x[i + 1] = v.y; }
    double beta, tau, scale;
    householder<1>(x[J], xs[QP_SEL], beta, tau, scale);
    const double st = -scale * tau;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        double dt[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = J + 1; i < QP_SEL; ++i) dt[i & 3] = fma(x[i], y[q][i], dt[i & 3]);
        const double dot = (dt[0] + dt[1]) + (dt[2] + dt[3]);
        const double cf = st * fma(scale, dot, y[q][J]);             // y_i -= v_i * tau * (v^T y), v_i = x_i * scale
#pragma unroll
        for (int i = J + 1; i < QP_SEL; ++i) y[q][i] = fma(x[i], cf, y[q][i]);
        if (t + 64 * NW * q == pcol) live &= ~(1u << q);
    }
    if (lane == 0 && wave == 0) sh.sel[J] = pcol;
}

// ---- panel: Householder step J on the 16 selected columns ----
// a[b][16]: rows 64 NW b + 64 wave + 16 g + idx of column sel[c] (b < CPL: a lane owns one 16-row block in each "layer" of 64 NW rows);
// diag bit b: that block is the panel's own rows k .. k + 15; rows above the panel (r < k) hold zeros.  x = column J below row k + J:
// all sources are the registers a[][] themselves (row_newbcast:J picks lane J of the 16-lane row); the rows <= k + J of the diagonal
// block are kept out by summing them separately (dots) / by a zero factor (update).
// The reflector tails stay UNSCALED in the registers (v_c = x_c * myscale below the diagonal, myscale kept by the lanes of column c,
// applied at the write-out): no per-step scaling pass, and v_c^T v_J = myscale_c (x_c[k+J] + scale_J x_c^T x_J) comes out of the same dots.
template <int J, int NW, int CPL>
__device__ __forceinline__ void panel_step(double (&a)[CPL][16], double (&trow)[QP_B], double& myscale, unsigned diag, unsigned blk_live, int c, int g, int wave,
                                           PanelShared<NW>& sh, double* tau_out) {
    constexpr int par = J & 1;
    if (blk_live) {
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < CPL; ++b) {
            if ((blk_live >> b) & 1u) {                                // wave-uniform: blocks above the panel are skipped
                double dlo0 = 0.0, dlo1 = 0.0, dhi0 = 0.0, dhi1 = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i <= J) { if (i & 1) dpp_fmac_self<J>(dlo1, a[b][i]); else dpp_fmac_self<J>(dlo0, a[b][i]); }
                    else { if (i & 1) dpp_fmac_self<J>(dhi1, a[b][i]); else dpp_fmac_self<J>(dhi0, a[b][i]); }
                }
                acc += (dhi0 + dhi1) + (((diag >> b) & 1u) ? 0.0 : dlo0 + dlo1);
                if ((diag >> b) & 1u) sh.prow[par][c] = a[b][J];
            }
        }
        const double part = rows4_sum(acc);
        if (g == 0) sh.part[par][c][wave] = part;
    } else if (g == 0) sh.part[par][c][wave] = 0.0;
    lds_barrier();
    double s_c = sh.part[par][c][0], s_j = sh.part[par][J][0];
#pragma unroll
    for (int q = 1; q < NW; ++q) { s_c += sh.part[par][c][q]; s_j += sh.part[par][J][q]; }
    const double alpha = sh.prow[par][J], apc = sh.prow[par][c];
    double beta, tau, scale;
    householder<2>(alpha, s_j, beta, tau, scale);
    const double zc = fma(scale, s_c, apc);                           // c > J: v_J^T a_c; c < J: v_J^T x_c (x_c = the unscaled tail of v_c)
    if (c == J) myscale = scale;
    if (blk_live) {
        const double mcf = c > J ? -scale * tau * zc : 0.0;
#pragma unroll
        for (int b = 0; b < CPL; ++b) {
            if ((blk_live >> b) & 1u) {
                const bool isd = (diag >> b) & 1u;
                const double mlo = isd ? 0.0 : mcf;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i <= J) dpp_axpy_self<J>(a[b][i], mlo);
                    else dpp_axpy_self<J>(a[b][i], mcf);
                }
                const double newd = c > J ? fma(-tau, zc, a[b][J]) : beta;   // row k + J: v = 1
                a[b][J] = (isd && c >= J) ? newd : a[b][J];
                dpp_fence(a[b]);
            }
        }
    }
    // compact WY: T[0:J, J] = -tau T[0:J, 0:J] (V^T v_J), T[J][J] = tau; lane c keeps row c of T
    if (wave == 0) {
        double zt = myscale * zc;                                     // (V^T v_J)_c for c < J
        double tj = 0.0;
        dpp_fence1(zt);
        t_row_dot(tj, trow, zt, std::make_integer_sequence<int, J>{});
        trow[J] = c < J ? -tau * tj : (c == J ? tau : 0.0);
        if (g == 0 && c == J) tau_out[J] = tau;
    }
}

}  // namespace

// One workgroup of NW waves per chain (64 NW CPL >= n: a lane owns CPL columns in the selection and CPL 16-row blocks of one column in
// the panel): selects the 16 columns of the panel that starts at step k and factors them.
template <int NW, int CPL>
__global__ __launch_bounds__(64 * NW) void qp_panel_kernel(Mat Am, QrWork w, int n, int k) {
    __shared__ PanelShared<NW> sh;
    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    double* __restrict__ pw = w.pw + (long)chain * w.pw_stride;
    const double* __restrict__ Y = pw;                         // [QP_SR][n]
    double* __restrict__ Vp = pw + (long)QP_SR * n + (long)n * k;                 // [n][QP_B] column-major clean copy of THIS panel's reflectors (all panels are kept: qp_formq_kernel)
    double* __restrict__ Tm = pw + (long)QP_SR * n + (long)n * n + (long)QP_B * k;   // [QP_B][QP_B] column-major, one per panel
    double* __restrict__ VTp = pw + (long)QP_SR * n + (long)n * n + (long)QP_B * n + (long)n * k;   // the same panel row-major ([n][QP_B]): the operand of V^T A is read along its rows
    int* __restrict__ pivpos = w.pivpos + (long)chain * w.pivpos_stride;
    double* tau = w.tau + (long)chain * w.tau_stride;
    int* jpvt = w.jpvt + (long)chain * w.jpvt_stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

#ifdef DQ_QP_STAMPS
    __shared__ unsigned long long qst[40];
#endif
    QST(0)
    // ---- selection ----
    {
        unsigned live = 0u;
        double y[CPL][QP_SEL]; int pp[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {                              // one round of loads: clamped addresses, masked afterwards
            const int col = t + 64 * NW * q, cc = col < n ? col : n - 1;
#pragma unroll
            for (int i = 0; i < QP_SEL; ++i) y[q][i] = Y[(long)i * n + cc];
            pp[q] = pivpos[cc];
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int col = t + 64 * NW * q;
            if (col < n && pp[q] < 0) live |= 1u << q;
            if (col >= n) {
#pragma unroll
                for (int i = 0; i < QP_SEL; ++i) y[q][i] = 0.0;
            }
        }
        if (y[0][0] == 1.2345e300) live = 0u;                            // (stamps: forces the loads to have landed before the first stamp)
        QST(1)
#define QP_SEL_STEP(J) select_step<J, NW, CPL>(y, live, t, lane, wave, sh); QST(2 + J)
        QP_SEL_STEP(0) QP_SEL_STEP(1) QP_SEL_STEP(2) QP_SEL_STEP(3) QP_SEL_STEP(4) QP_SEL_STEP(5) QP_SEL_STEP(6) QP_SEL_STEP(7)
        QP_SEL_STEP(8) QP_SEL_STEP(9) QP_SEL_STEP(10) QP_SEL_STEP(11) QP_SEL_STEP(12) QP_SEL_STEP(13) QP_SEL_STEP(14) QP_SEL_STEP(15)
#undef QP_SEL_STEP
    }
    lds_barrier();
    // ---- panel ----
    const int c = lane & 15, g = lane >> 4;
    const int mycol = sh.sel[c];
    double a[CPL][16], trow[QP_B], myscale = 0.0;
    unsigned diag = 0u, blk_live = 0u;
#pragma unroll
    for (int b = 0; b < CPL; ++b) {
        const int r0 = 64 * NW * b + 64 * wave + 16 * g;
        if (r0 == k) diag |= 1u << b;
        if (64 * NW * b + 64 * wave + 64 > k && 64 * NW * b + 64 * wave < n) blk_live |= 1u << b;   // wave-uniform: some row of the wave's 64 is in the panel
        const double2* src = reinterpret_cast<const double2*>(A + (long)n * mycol + r0);      // 16 contiguous rows of one column: n and r0 are multiples of 16, so 16-byte pieces
        const bool rows_live = r0 >= k && r0 < n;
#pragma unroll
        for (int i = 0; i < 16; i += 2) { const double2 v = rows_live ? src[i >> 1] : double2{0.0, 0.0}; a[b][i] = v.x; a[b][i + 1] = v.y; }
        dpp_fence(a[b]);
    }
#pragma unroll
    for (int i = 0; i < QP_B; ++i) trow[i] = 0.0;
    double* tau_out = tau + k;
#ifdef DQ_QP_STAMPS
    if (a[0][0] == 1.2345e300) myscale = 1.0;
#endif
    QST(18)
#define QP_PAN(J) panel_step<J, NW, CPL>(a, trow, myscale, diag, blk_live, c, g, wave, sh, tau_out); QST(19 + J)
    QP_PAN(0) QP_PAN(1) QP_PAN(2) QP_PAN(3) QP_PAN(4) QP_PAN(5) QP_PAN(6) QP_PAN(7)
    QP_PAN(8) QP_PAN(9) QP_PAN(10) QP_PAN(11) QP_PAN(12) QP_PAN(13) QP_PAN(14) QP_PAN(15)
#undef QP_PAN
    // ---- write-out: R0 / beta / reflectors in place, the clean reflector panel (unit diagonal, zeros above), T, jpvt, pivpos ----
#pragma unroll
    for (int b = 0; b < CPL; ++b) {
        const int r0 = 64 * NW * b + 64 * wave + 16 * g;
        if (r0 >= k && r0 < n) {
            const bool isd = (diag >> b) & 1u;
            double2* dst = reinterpret_cast<double2*>(A + (long)n * mycol + r0);
            double2* vdst = reinterpret_cast<double2*>(Vp + (long)n * c + r0);
            double v[16], vc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool below = !isd || i > c;                         // strictly below the column's diagonal: the reflector tail
                v[i] = below ? a[b][i] * myscale : a[b][i];
                vc[i] = below ? v[i] : (i == c ? 1.0 : 0.0);
                VTp[(long)QP_B * (r0 + i) + c] = vc[i];
            }
#pragma unroll
            for (int i = 0; i < 16; i += 2) { dst[i >> 1] = double2{v[i], v[i + 1]}; vdst[i >> 1] = double2{vc[i], vc[i + 1]}; }
        }
    }
    if (wave == 0 && g == 0) {
#pragma unroll
        for (int i = 0; i < QP_B; ++i) Tm[c + QP_B * i] = trow[i];
        jpvt[k + c] = mycol;
        pivpos[mycol] = k + c;
    }
#ifdef DQ_QP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QST(35)
    __syncthreads();
    if (t == 0 && k == 0 && blockIdx.y == 0) {
        printf("panel k=0 n=%d (100 MHz ticks x 24 = clk @2.4GHz): load Y %llu | sel steps", n, qst[1] - qst[0]);
        for (int j = 0; j < 16; ++j) printf(" %llu", qst[2 + j] - qst[1 + j]);
        printf(" | gather %llu | panel steps", qst[18] - qst[17]);
        for (int j = 0; j < 16; ++j) printf(" %llu", qst[19 + j] - qst[18 + j]);
        printf(" | write-out %llu | total %llu\n", qst[35] - qst[34], qst[35] - qst[0]);
    }
#endif
}

// grid.x = n / 16 column blocks, QP_UW waves each.  UPDATE: A[k:, cols] <- (I - V T V^T)^T A[k:, cols] for the live columns of the block, then
// the sketch of rows >= k + 16 of the result; !UPDATE (before the first panel): pivpos = -1 and the sketch of A itself.
// A wave keeps its row tiles (tile rt = wave + QP_UW ti, ti < TPW) in registers in the accumulator layout from the first load on: the same
// registers are the B operand of W = V^T A (k-step s <-> row kk + 4 s of the tile: any partition of k works as long as V is fed the same
// way), the accumulator of A - V W', and the B operand of the sketch product.
constexpr int QP_UW = 8;      // waves per workgroup of the update / form-Q kernels (4: 6.7 / 45.9 us at n = 256 against 6.3 / 40.9; 16: 8.3 / 166)
template <bool UPDATE, int TPW>
__global__ __launch_bounds__(64 * QP_UW) void qp_update_kernel(Mat Am, QrWork w, int n, int k) {
    constexpr int YT = QP_SR / 16;
    __shared__ double red[QP_UW][YT * 4][64];                  // per wave partial tiles (W: 4 registers, Y: 4 YT registers)
    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    double* __restrict__ pw = w.pw + (long)chain * w.pw_stride;
    double* __restrict__ Y = pw;
    const double* __restrict__ Vp = pw + (long)QP_SR * n + (long)n * k;
    const double* __restrict__ Tm = pw + (long)QP_SR * n + (long)n * n + (long)QP_B * k;
    const double* __restrict__ VTp = pw + (long)QP_SR * n + (long)n * n + (long)QP_B * n + (long)n * k;
    int* __restrict__ pivpos = w.pivpos + (long)chain * w.pivpos_stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int col = 16 * blockIdx.x + r16;
    double* __restrict__ Ac = A + (long)n * col;
    const int m_tiles = (n - k) / 16;
    // every global load of the launch is issued here, before the first use: the operands were written by the previous kernel (another
    // CU, mostly another XCD), so each dependent round of loads costs a cold miss (~2 us) -- one round instead of three
    int pp = -1;
    if (UPDATE) pp = pivpos[col];
    d4 At[TPW]; double v1[TPW][4], v2[TPW][4], tv[4];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int rt = wave + QP_UW * ti;
        At[ti] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) { v1[ti][s] = 0.0; v2[ti][s] = 0.0; }
        if (rt < m_tiles) {
            const int r0 = k + 16 * rt;
#pragma unroll
            for (int r = 0; r < 4; ++r) At[ti][r] = Ac[r0 + kk + 4 * r];
            if (UPDATE) {
#pragma unroll
                for (int s = 0; s < 4; ++s) { v1[ti][s] = VTp[(long)QP_B * (r0 + kk + 4 * s) + r16]; v2[ti][s] = Vp[(long)n * (kk + 4 * s) + (r0 + r16)]; }   // both along the lanes' fast index
            }
        }
    }
    if (UPDATE) {
#pragma unroll
        for (int s = 0; s < 4; ++s) tv[s] = Tm[(kk + 4 * s) + QP_B * r16];
    }
    const bool live_col = pp < 0;
    if (UPDATE) { if (__ballot(live_col) == 0ULL) return; }   // every wave sees the same 16 columns: a uniform exit
    else if (t < 16) pivpos[col] = -1;
    d4 wp = {0.0, 0.0, 0.0, 0.0};
    if (UPDATE) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v1[ti][s], At[ti][s], acc, 0, 0, 0);     // tiles beyond the matrix hold zeros
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
        __syncthreads();
        d4 W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = red[0][r][lane];
#pragma unroll
            for (int q = 1; q < QP_UW; ++q) v += red[q][r][lane];
            W[r] = live_col ? v : 0.0;
        }
        // W' = T^T W: a = T[i = kk + 4 s][i' = r16], b = W[kk + 4 s][c] (the accumulator registers as they are)
#pragma unroll
        for (int s = 0; s < 4; ++s) wp = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[s], W[s], wp, 0, 0, 0);
        wp = -wp;                                                 // A - V W'
        __syncthreads();
    }
    d4 ya[YT];
#pragma unroll
    for (int yt = 0; yt < YT; ++yt) ya[yt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int rt = wave + QP_UW * ti;
        if (rt < m_tiles) {
            const int r0 = k + 16 * rt;
            if (UPDATE) {
#pragma unroll
                for (int s = 0; s < 4; ++s) At[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(v2[ti][s], wp[s], At[ti], 0, 0, 0);
                if (live_col) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Ac[r0 + kk + 4 * r] = At[ti][r];
                }
            }
            if (!UPDATE || rt >= 1) {                              // rows of the NEXT trailing matrix: sketch them while they are in registers
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned bits = qp_row_bits((unsigned)(r0 + kk + 4 * s));
#pragma unroll
                    for (int yt = 0; yt < YT; ++yt) ya[yt] = __builtin_amdgcn_mfma_f64_16x16x4f64(qp_sign(bits, 16 * yt + r16), At[ti][s], ya[yt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int yt = 0; yt < YT; ++yt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * yt + r][lane] = ya[yt][r];
    __syncthreads();
    for (int idx = t; idx < YT * 4 * 64; idx += 64 * QP_UW) {
        const int l = idx & 63, q = idx >> 6, yt = q >> 2, r = q & 3;
        double v = red[0][q][l];
#pragma unroll
        for (int qq = 1; qq < QP_UW; ++qq) v += red[qq][q][l];
        Y[(long)(16 * yt + (l >> 4) + 4 * r) * n + 16 * blockIdx.x + (l & 15)] = v;
    }
}

// Q = H_0 H_1 ... H_{n/16 - 1} applied to I, every factor in its compact-WY form I - V_k T_k V_k^T (the clean reflector panels and the
// T factors the panel kernel left behind): grid.x = n / 16 column blocks of Q, each kept in registers as accumulator tiles (tile rt =
// wave + QP_UW ti) through all the factors that touch it -- columns 16 cb .. 16 cb + 15 of the identity are unchanged by the panels
// k > cb (their reflectors start below row 16 k), so block cb applies panels k = cb .. 0 -- two MFMA products and one LDS reduction
// per factor, the next factor's operands requested while the current one is multiplied.  77 us (formq_blocked_kernel: one column per
// 16-lane row, reflector by reflector) -> the figure in DESIGN at n = 256.
template <int TPW>
__global__ __launch_bounds__(64 * QP_UW) void qp_formq_kernel(QrWork w, Mat Lm, int n) {
    __shared__ double red[QP_UW][4][64];
    const int chain = blockIdx.y;
    const double* __restrict__ pw = w.pw + (long)chain * w.pw_stride;
    const double* __restrict__ Vall = pw + (long)QP_SR * n;
    const double* __restrict__ Tall = Vall + (long)n * n;
    const double* __restrict__ VTall = Tall + (long)QP_B * n;
    double* __restrict__ L = Lm.at(chain);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int cb = blockIdx.x, col = 16 * cb + r16;
    const int n_tiles = n / 16;
    d4 Qt[TPW];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int rt = wave + QP_UW * ti;
#pragma unroll
        for (int r = 0; r < 4; ++r) Qt[ti][r] = (16 * rt + kk + 4 * r == col) ? 1.0 : 0.0;
    }
    double v1[TPW][4], v2[TPW][4], tv[4], n1[TPW][4], n2[TPW][4], nt_[4];
    auto load = [&](int k, double (&a1)[TPW][4], double (&a2)[TPW][4], double (&at)[4]) {
        const double* Vp = Vall + (long)n * 16 * k;
        const double* Tm = Tall + (long)QP_B * 16 * k;
        const double* VTp = VTall + (long)n * 16 * k;
#pragma unroll
        for (int s = 0; s < 4; ++s) at[s] = Tm[r16 + QP_B * (kk + 4 * s)];          // W' = T W: a = T[i' = r16][i = kk + 4 s]
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            const int rt = wave + QP_UW * ti;
            const bool in = rt < n_tiles && rt >= k;                                // rows of the factor: 16 k .. n - 1
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a1[ti][s] = in ? VTp[(long)QP_B * (16 * rt + kk + 4 * s) + r16] : 0.0;
                a2[ti][s] = in ? Vp[(long)n * (kk + 4 * s) + (16 * rt + r16)] : 0.0;
            }
        }
    };
    load(cb, v1, v2, tv);
    for (int k = cb; k >= 0; --k) {
        if (k > 0) load(k - 1, n1, n2, nt_);
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v1[ti][s], Qt[ti][s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
        __syncthreads();
        d4 W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = red[0][r][lane];
#pragma unroll
            for (int q = 1; q < QP_UW; ++q) v += red[q][r][lane];
            W[r] = v;
        }
        d4 wp = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) wp = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[s], W[s], wp, 0, 0, 0);
        wp = -wp;
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
            for (int s = 0; s < 4; ++s) Qt[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(v2[ti][s], wp[s], Qt[ti], 0, 0, 0);   // v2 = 0 above the factor's rows
        __syncthreads();
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
            for (int s = 0; s < 4; ++s) { v1[ti][s] = n1[ti][s]; v2[ti][s] = n2[ti][s]; }
#pragma unroll
        for (int s = 0; s < 4; ++s) tv[s] = nt_[s];
    }
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int rt = wave + QP_UW * ti;
        if (rt < n_tiles) {
#pragma unroll
            for (int r = 0; r < 4; ++r) L[(long)n * col + 16 * rt + kk + 4 * r] = Qt[ti][r];
        }
    }
}

int launch_qr_panel_formq(QrWork w, Mat L, int n, int n_chains, hipStream_t s) {
    const dim3 grid(n / 16, n_chains);
    const int tpw = (n / 16 + QP_UW - 1) / QP_UW;
    if (tpw <= 2) hipLaunchKernelGGL((qp_formq_kernel<2>), grid, dim3(64 * QP_UW), 0, s, w, L, n);
    else if (tpw <= 5) hipLaunchKernelGGL((qp_formq_kernel<5>), grid, dim3(64 * QP_UW), 0, s, w, L, n);
    else hipLaunchKernelGGL((qp_formq_kernel<8>), grid, dim3(64 * QP_UW), 0, s, w, L, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

bool qr_panel_ok(int n, const QrWork& w) { return n >= 16 && n <= 1024 && n % 16 == 0 && w.pw != nullptr && w.pivpos != nullptr && w.pw_stride >= qr_panel_work_doubles(n); }
long qr_panel_work_doubles(int n) { return (long)(QP_SR + 2 * n) * n + (long)QP_B * n; }   // Y | V panels | T factors | V panels row-major

int launch_qr_panel(Mat A, QrWork w, int n, int n_chains, hipStream_t s) {
    if (!qr_panel_ok(n, w)) { set_error("panel QR: n must be a multiple of 16 in [16, 1024] and the workspace present"); return -1; }
    const dim3 ugrid(n / 16, n_chains);
    const int tpw = (n / 16 + QP_UW - 1) / QP_UW;
#define QP_UPD(U, K) do { if (tpw <= 2) hipLaunchKernelGGL((qp_update_kernel<U, 2>), ugrid, dim3(64 * QP_UW), 0, s, A, w, n, K); \
                          else if (tpw <= 5) hipLaunchKernelGGL((qp_update_kernel<U, 5>), ugrid, dim3(64 * QP_UW), 0, s, A, w, n, K); \
                          else hipLaunchKernelGGL((qp_update_kernel<U, 8>), ugrid, dim3(64 * QP_UW), 0, s, A, w, n, K); } while (0)
    QP_UPD(false, 0);
    for (int k = 0; k < n; k += QP_B) {
#define QP_LAUNCH(NW, CPL) hipLaunchKernelGGL((qp_panel_kernel<NW, CPL>), dim3(1, n_chains), dim3(64 * NW), 0, s, A, w, n, k)
        if (n <= 64) QP_LAUNCH(1, 1); else if (n <= 128) QP_LAUNCH(2, 1); else if (n <= 256) QP_LAUNCH(4, 1); else if (n <= 512) QP_LAUNCH(4, 2);
        else if (n <= 576) QP_LAUNCH(3, 3); else if (n <= 768) QP_LAUNCH(4, 3); else QP_LAUNCH(4, 4);
#undef QP_LAUNCH
        if (k + QP_B < n) QP_UPD(true, k);
    }
#undef QP_UPD
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
