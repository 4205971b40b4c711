// qr.hip -- stablelinalg::to_LDR (source/stablelinalg.cpp:35-55) on device:
// Householder QR with column pivoting (the dgeqp3 behind arma::qr(...,"vector")),
// explicit Q (dorgqr), d = |diag R0| and R = diag(1/d) R0 P^T.
//
//   qrcp_kernel   one 1024-thread workgroup per chain.  LAPACK dlaqp2
//                 semantics: pivot = remaining column of largest partial norm,
//                 Householder reflector from dlarfg (beta = -sign(alpha)*norm),
//                 partial norms down-dated and recomputed on cancellation.
//                 Columns are NOT swapped: step i's reflector and R0 column
//                 stay in the pivot column's own storage, jpvt[i] records it.
//                 Wave w owns columns w, w+16, ...; a lane owns rows lane,
//                 lane+64, ...; a wave streams its live columns from L2 in
//                 batches of 8 (all loads in flight before the first use), the
//                 per-column dot is a wave reduction.  While updating, each
//                 wave keeps the register image of its best (largest-norm)
//                 column and parks it in LDS, so the next step's pivot column
//                 is read from LDS, not from L2: the serial part of a step
//                 touches no global memory.
//   formq_kernel  Q = H_0 ... H_{n-1} applied to I.  Column tiles of Q are
//                 independent, so the grid is (n/16 tiles) x chains and every
//                 wave keeps its 4 columns in registers: no LDS, no barriers.
//   assemble_r_kernel  d and the row-normalised R in original column order.
#include "common.h"
#include "wave.h"
#include <cstdlib>

#define DQ_TRY_RC(expr) do { int _rc = (expr); if (_rc != 0) return _rc; } while (0)

namespace dq {


// NR = rows per lane (n <= 64*NR)
template <int NR>
__global__ __launch_bounds__(1024) void qrcp_kernel(Mat Am, QrWork w, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* vn1 = reinterpret_cast<double*>(smem);          // [n]
    double* vn2 = vn1 + n;                                  // [n]
    double* v = vn2 + n;                                    // [n]   current Householder vector
    double* candcol = v + n;                                // [16][n] best column of each wave
    double* cand_v = candcol + 16 * (size_t)n;              // [16]
    double* red = cand_v + 16;                              // [16]
    int* cand_i = reinterpret_cast<int*>(red + 16);         // [16]
    int* pivpos = cand_i + 16;                              // [n]  -1 = live, else pivot position
    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    double* tau = w.tau + (long)chain * w.tau_stride;
    int* jpvt = w.jpvt + (long)chain * w.jpvt_stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double tol3z = 1.0536712127723509e-08;    // sqrt(dlamch('Epsilon')) = sqrt(2^-53)
    constexpr int QR_CB = NR <= 2 ? 8 : (NR <= 4 ? 4 : (NR <= 9 ? 2 : 1));   // columns a wave keeps in flight (register budget: 128 VGPRs)

    // initial column norms + each wave's best column
    {
        double bestn = -1.0; int besti = -1; double bcol[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) bcol[q] = 0.0;
        for (int c = wave; c < n; c += 16) {
            double a[NR]; double s = 0.0;
#pragma unroll
            for (int q = 0; q < NR; ++q) { const int r = lane + 64 * q; a[q] = (r < n) ? A[r + (long)n * c] : 0.0; s += a[q] * a[q]; }
            s = wave_sum(s);
            const double nn = sqrt(s);
            if (lane == 0) { vn1[c] = nn; vn2[c] = nn; pivpos[c] = -1; }
            if (nn > bestn) {
                bestn = nn; besti = c;
#pragma unroll
                for (int q = 0; q < NR; ++q) bcol[q] = a[q];
            }
        }
        if (lane == 0) { cand_v[wave] = bestn; cand_i[wave] = besti; }
#pragma unroll
        for (int q = 0; q < NR; ++q) { const int r = lane + 64 * q; if (r < n) candcol[(size_t)wave * n + r] = bcol[q]; }
    }
    __syncthreads();

    for (int i = 0; i < n; ++i) {
        // (a) pivot among the 16 wave candidates: largest norm, lowest column index on ties
        double best = cand_v[0]; int bw = 0; int p = cand_i[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const double ob = cand_v[k]; const int oi = cand_i[k];
            if (oi >= 0 && (p < 0 || ob > best || (ob == best && oi < p))) { best = ob; bw = k; p = oi; }
        }
        // (b) Householder vector from the parked image of column p (dlarfg)
        const double x = (t >= i && t < n) ? candcol[(size_t)bw * n + t] : 0.0;
        double ss = (t > i) ? x * x : 0.0;
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        double xnorm2 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) xnorm2 += red[k];
        const double al = candcol[(size_t)bw * n + i];
        double tau_i = 0.0, beta = al, scale = 0.0;
        if (xnorm2 != 0.0) {
            beta = -copysign(sqrt(al * al + xnorm2), al);
            tau_i = (beta - al) / beta;
            scale = 1.0 / (al - beta);
        }
        if (t < n) {
            double vr = 0.0;
            if (t == i) { vr = 1.0; A[t + (long)n * p] = beta; }
            else if (t > i) { vr = x * scale; A[t + (long)n * p] = vr; }
            v[t] = vr;
        }
        if (t == 0) { tau[i] = tau_i; jpvt[i] = p; pivpos[p] = i; }
        __syncthreads();
        // (c) apply H to the live columns of this wave, down-date norms, track the wave's best column
        double vr[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) { const int r = lane + 64 * q; vr[q] = (r < n) ? v[r] : 0.0; }
        double bestn = -1.0; int besti = -1; double bcol[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) bcol[q] = 0.0;
        for (int cb = wave; cb < n; cb += 16 * QR_CB) {
            double a[QR_CB][NR]; bool live[QR_CB];
#pragma unroll
            for (int k = 0; k < QR_CB; ++k) {
                const int c = cb + 16 * k;
                live[k] = c < n && pivpos[c] < 0;
#pragma unroll
                for (int q = 0; q < NR; ++q) { const int r = lane + 64 * q; a[k][q] = (live[k] && r >= i && r < n) ? A[r + (long)n * c] : 0.0; }
            }
#pragma unroll
            for (int k = 0; k < QR_CB; ++k) {
                if (!live[k]) continue;
                const int c = cb + 16 * k;
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < NR; ++q) s += a[k][q] * vr[q];
                s = wave_sum(s) * tau_i;
                double tail = 0.0, ai_q = 0.0;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const int r = lane + 64 * q;
                    if (r >= i && r < n) {
                        a[k][q] -= s * vr[q];
                        if (tau_i != 0.0) A[r + (long)n * c] = a[k][q];
                        if (r > i) tail += a[k][q] * a[k][q];
                    }
                    if (q == (i >> 6)) ai_q = a[k][q];
                }
                const double aic = readlane_f64(ai_q, i & 63);   // A[i, c]
                double n1 = vn1[c];
                if (n1 != 0.0) {                         // norm down-date (dlaqp2)
                    double temp = fabs(aic) / n1; temp = fmax(0.0, 1.0 - temp * temp);
                    const double rr = n1 / vn2[c];
                    const double temp2 = temp * rr * rr;
                    if (temp2 <= tol3z) {
                        tail = wave_sum(tail);
                        n1 = (i + 1 < n) ? sqrt(tail) : 0.0;
                        if (lane == 0) { vn1[c] = n1; vn2[c] = n1; }
                    } else {
                        n1 = n1 * sqrt(temp);
                        if (lane == 0) vn1[c] = n1;
                    }
                }
                if (n1 > bestn || besti < 0) {
                    bestn = n1; besti = c;
#pragma unroll
                    for (int q = 0; q < NR; ++q) bcol[q] = a[k][q];
                }
            }
        }
        if (lane == 0) { cand_v[wave] = bestn; cand_i[wave] = besti; }
        if (besti >= 0) {
#pragma unroll
            for (int q = 0; q < NR; ++q) { const int r = lane + 64 * q; if (r < n) candcol[(size_t)wave * n + r] = bcol[q]; }
        }
        __syncthreads();
    }
}

// Q = H_0 ... H_{n-1} I, one COLUMN of Q per 16-lane DPP row: lane = 16*cq + rg holds rows 16*j + rg
// (j < NRR) of column c0 + cq in registers, so the dot product of a reflector with a column is NRR FMAs
// plus a 4-step DPP row reduction, and the four columns of a wave are processed at once (a full
// 64-lane wave_sum per column made this kernel 6x slower).  Reflector i lives in column jpvt[i] of A
// (rows > i); the next reflector is prefetched while the current one is applied.  Column tiles of Q are
// independent: grid = (n/16 tiles) x chains, no LDS, no barriers.
template <int NRR>
__global__ __launch_bounds__(256) void formq_kernel(CMat Am, const double* tau_p, long tau_stride, const int* jpvt_p, long jpvt_stride, Mat Qm, int n) {
    const int chain = blockIdx.y;
    const double* __restrict__ A = Am.at(chain);
    const double* __restrict__ tau = tau_p + (long)chain * tau_stride;
    const int* __restrict__ jpvt = jpvt_p + (long)chain * jpvt_stride;
    double* __restrict__ Q = Qm.at(chain);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = lane & 15, cq = lane >> 4;
    const int c = blockIdx.x * 16 + wave * 4 + cq;             // my column
    const int cw_max = min(blockIdx.x * 16 + wave * 4 + 3, n - 1);   // largest column of this wave
    const bool wave_idle = blockIdx.x * 16 + wave * 4 >= n;
    double q[NRR];
#pragma unroll
    for (int j = 0; j < NRR; ++j) q[j] = (16 * j + rg == c) ? 1.0 : 0.0;
    // three reflectors in flight; jpvt / tau are staged in LDS first: read from HBM inside the loop, the pivot
    // index is a dependent load in front of every column fetch (one L2 latency per reflector whatever the depth)
    __shared__ int s_jpvt[1024];
    __shared__ double s_tau[1024];
    for (int k = threadIdx.x; k < n; k += blockDim.x) { s_jpvt[k] = jpvt[k]; s_tau[k] = tau[k]; }
    __syncthreads();
    if (wave_idle) return;
    constexpr int PD = 3;
    double vq[PD][NRR]; double tq[PD];
    auto fetch = [&](int slot, int ii) {
        const int ic = ii > 0 ? ii : 0;                      // clamped: harmless reload past the end
        const long cb = (long)n * s_jpvt[ic]; tq[slot] = s_tau[ic];
#pragma unroll
        for (int j = 0; j < NRR; ++j) vq[slot][j] = A[min(16 * j + rg, n - 1) + cb];      // raw: masked where it is used (see FormQBlock::fetch)
    };
    auto apply = [&](int slot, int i) {
        const double ti = tq[slot];
        if (ti == 0.0) return;
        double v[NRR];
#pragma unroll
        for (int j = 0; j < NRR; ++j) { const int r = 16 * j + rg; v[j] = (r > i && r < n) ? vq[slot][j] : (r == i ? 1.0 : 0.0); }
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NRR; ++j) s += v[j] * q[j];
        s = row16_sum(s) * ti;
        if (c < i) s = 0.0;                                  // column still e_c: untouched by H_i
#pragma unroll
        for (int j = 0; j < NRR; ++j) q[j] -= s * v[j];
    };
    fetch(0, cw_max); fetch(1, cw_max - 1); fetch(2, cw_max - 2);
    int i = cw_max;
    for (; i >= 2; i -= 3) {                                 // static slot indices: the ring is unrolled by hand
        apply(0, i);     fetch(0, i - 3);
        apply(1, i - 1); fetch(1, i - 4);
        apply(2, i - 2); fetch(2, i - 5);
    }
    if (i >= 0) apply(0, i);
    if (i >= 1) apply(1, i - 1);
    if (c < n) {
#pragma unroll
        for (int j = 0; j < NRR; ++j) { const int r = 16 * j + rg; if (r < n) Q[r + (long)n * c] = q[j]; }
    }
}

// formq for n = 16 * NRR exactly (n = 16, 32, 64, 128, 256): the reflectors are walked in blocks of 16 (block M = reflectors 16M .. 16M+15),
// one code variant per block, so that register indices stay static while
//   * rows below the block (j < M) are neither loaded nor multiplied: H_i only touches rows >= i -- half the work on average,
//   * only the row groups j = M - 1, M need the "rows <= i hold R, row i is the implicit 1" mask; j > M is loaded as it is.
// The ring of four prefetched reflectors runs across block boundaries (a block's fetches include row group M - 1 for that).
// The dot product uses two accumulators (a single chain of dependent fp64 FMAs costs its full latency per term).
template <int NRR, int M>
struct FormQBlock {
    static constexpr int J0 = M > 0 ? M - 1 : 0;
    // raw loads only: a select on a value just loaded makes the wave wait for that load inside the fetch, i.e. one memory latency per
    // reflector whatever the depth of the ring (measured: 77 us per 256 x 256 form-Q with the mask here); the
    // "rows <= i hold R, row i is the implicit 1" mask is applied where the value is used
    static __device__ __forceinline__ void fetch(double (&v)[NRR], double& t, const double* __restrict__ A, const int* s_jpvt, const double* s_tau,
                                                 int n, int rg, int ii) {
        const int ic = ii > 0 ? ii : 0;                      // clamped: harmless reload past the end
        const long cb = (long)n * s_jpvt[ic]; t = s_tau[ic];
#pragma unroll
        for (int j = J0; j < NRR; ++j) v[j] = A[16 * j + rg + cb];
    }
    static __device__ __forceinline__ void apply(double (&q)[NRR], const double (&v)[NRR], double ti, int c, int i, int rg) {
        if (ti == 0.0) return;
        const int rm = 16 * M + rg;                          // row group M is the only one of j >= M that reaches up to row i
        const double vm = (rm > i) ? v[M] : (rm == i ? 1.0 : 0.0);
        double s0 = vm * q[M], s1 = 0.0;
#pragma unroll
        for (int j = M + 1; j < NRR; ++j) { if ((j - M) & 1) s1 = fma(v[j], q[j], s1); else s0 = fma(v[j], q[j], s0); }
        double s = row16_sum(s0 + s1) * ti;
        if (c < i) s = 0.0;                                  // column still e_c: untouched by H_i
        q[M] = fma(-s, vm, q[M]);
#pragma unroll
        for (int j = M + 1; j < NRR; ++j) q[j] = fma(-s, v[j], q[j]);
    }
    static __device__ __forceinline__ void run(double (&q)[NRR], double (&vq)[4][NRR], double (&tq)[4], const double* __restrict__ A, const int* s_jpvt,
                                               const double* s_tau, int n, int rg, int c, int cw_max) {
        const int mtop = cw_max >> 4;
        if (M <= mtop) {                                     // wave-uniform
            const int hi = M == mtop ? cw_max : 16 * M + 15;
            if (M == mtop) {
#pragma unroll
                for (int e = 0; e < 4; ++e) fetch(vq[e], tq[e], A, s_jpvt, s_tau, n, rg, hi - e);
            }
            for (int top = hi; top >= 16 * M; top -= 4) {    // (hi + 1) is a multiple of 4: whole groups
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    apply(q, vq[e], tq[e], c, top - e, rg);
                    fetch(vq[e], tq[e], A, s_jpvt, s_tau, n, rg, top - e - 4);
                }
            }
        }
        FormQBlock<NRR, M - 1>::run(q, vq, tq, A, s_jpvt, s_tau, n, rg, c, cw_max);
    }
};
template <int NRR>
struct FormQBlock<NRR, -1> {
    static __device__ __forceinline__ void run(double (&)[NRR], double (&)[4][NRR], double (&)[4], const double*, const int*, const double*, int, int, int, int) {}
};
template <int NRR>
__global__ __launch_bounds__(256) void formq_blocked_kernel(CMat Am, const double* tau_p, long tau_stride, const int* jpvt_p, long jpvt_stride, Mat Qm, int n) {
    const int chain = blockIdx.y;
    const double* __restrict__ A = Am.at(chain);
    const double* __restrict__ tau = tau_p + (long)chain * tau_stride;
    const int* __restrict__ jpvt = jpvt_p + (long)chain * jpvt_stride;
    double* __restrict__ Q = Qm.at(chain);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = lane & 15, cq = lane >> 4;
    const int c = blockIdx.x * 16 + wave * 4 + cq;             // my column (< n: n is a multiple of 16)
    const int cw_max = blockIdx.x * 16 + wave * 4 + 3;         // largest column of this wave
    __shared__ int s_jpvt[16 * NRR];
    __shared__ double s_tau[16 * NRR];
    for (int k = threadIdx.x; k < n; k += blockDim.x) { s_jpvt[k] = jpvt[k]; s_tau[k] = tau[k]; }
    __syncthreads();
    double q[NRR];
#pragma unroll
    for (int j = 0; j < NRR; ++j) q[j] = (16 * j + rg == c) ? 1.0 : 0.0;
    double vq[4][NRR]; double tq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        tq[e] = 0.0;
#pragma unroll
        for (int j = 0; j < NRR; ++j) vq[e][j] = 0.0;
    }
    FormQBlock<NRR, NRR - 1>::run(q, vq, tq, A, s_jpvt, s_tau, n, rg, c, cw_max);
#pragma unroll
    for (int j = 0; j < NRR; ++j) { const int r = 16 * j + rg; if (r < n) Q[r + (long)n * c] = q[j]; }
}

static bool formq_blocked_ok(int n) {
    return (n == 16 || n == 32 || n == 64 || n == 128 || n == 256 || n == 576);      // NRR = n / 16 exactly: no row bound checks in the kernel
}
static void launch_formq_blocked(CMat A, const QrWork& w, Mat L, int n, int n_chains, hipStream_t s) {
    const dim3 grid(n / 16, n_chains), block(256);
#define DQ_FQ(NRR) hipLaunchKernelGGL((formq_blocked_kernel<NRR>), grid, block, 0, s, A, (const double*)w.tau, w.tau_stride, (const int*)w.jpvt, w.jpvt_stride, L, n)
    if (n <= 16) DQ_FQ(1); else if (n <= 32) DQ_FQ(2); else if (n <= 64) DQ_FQ(4); else if (n <= 128) DQ_FQ(8); else if (n <= 256) DQ_FQ(16); else DQ_FQ(36);
#undef DQ_FQ
}

// d[j] = |R0[j,j]|;  R[:, jpvt[j]] = R0[:, j] / d   (source/stablelinalg.cpp:47-52).
// Without column swaps R0[:, j] sits in column jpvt[j] of A, which is also its destination.
__global__ void assemble_r_kernel(CMat Am, const int* jpvt_p, long jpvt_stride, Vec dv, Mat Rm, int n) {
    const int chain = blockIdx.y;
    const double* __restrict__ A = Am.at(chain);
    const int* jpvt = jpvt_p + (long)chain * jpvt_stride;
    double* __restrict__ R = Rm.at(chain);
    double* d = dv.at(chain);
    const int j = blockIdx.x;                     // pivot position
    const int col = jpvt[j];
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        double val = 0.0;
        if (r <= j) val = A[r + (long)n * col] / fabs(A[r + (long)n * jpvt[r]]);
        R[r + (long)n * col] = val;
    }
    if (threadIdx.x == 0) d[j] = fabs(A[j + (long)n * col]);
}

int launch_qrcp_coop(Mat A, QrWork w, int n, int n_chains, hipStream_t s);     // qr_coop.hip
int launch_qrcp_colown(Mat A, QrWork w, int n, int n_chains, hipStream_t s);   // qr_colown.hip

// NRSEL = 0: on-chip QRCP (n <= 256: the single-CU column-owner kernel); -1: P cooperating workgroups (n > 256);
// otherwise the single-workgroup streaming kernel with NRSEL rows per lane
template <int NRSEL>
static int launch_to_ldr_nr(Mat A, Mat L, Vec d, Mat R, QrWork w, int n, int n_chains, hipStream_t s) {
    constexpr int NR = NRSEL <= 0 ? 4 : NRSEL;
    if (NRSEL <= 0) {
        if (NRSEL == -2) DQ_TRY_RC(launch_qr_panel(A, w, n, n_chains, s));
        else if (NRSEL == 0) DQ_TRY_RC(launch_qrcp_colown(A, w, n, n_chains, s));
        else DQ_TRY_RC(launch_qrcp_coop(A, w, n, n_chains, s));
        if (NRSEL == -2) DQ_TRY_RC(launch_qr_panel_formq(w, L, n, n_chains, s));
        else if (formq_blocked_ok(n)) launch_formq_blocked(CMat(A), w, L, n, n_chains, s);
        else if (n <= 256) hipLaunchKernelGGL((formq_kernel<16>), dim3((n + 15) / 16, n_chains), dim3(256), 0, s, CMat(A), (const double*)w.tau, w.tau_stride,
                                              (const int*)w.jpvt, w.jpvt_stride, L, n);
        else if (n <= 576) hipLaunchKernelGGL((formq_kernel<36>), dim3((n + 15) / 16, n_chains), dim3(256), 0, s, CMat(A), (const double*)w.tau, w.tau_stride,
                                              (const int*)w.jpvt, w.jpvt_stride, L, n);
        else hipLaunchKernelGGL((formq_kernel<64>), dim3((n + 15) / 16, n_chains), dim3(256), 0, s, CMat(A), (const double*)w.tau, w.tau_stride,
                                (const int*)w.jpvt, w.jpvt_stride, L, n);
        hipLaunchKernelGGL(assemble_r_kernel, dim3(n, n_chains), dim3(128), 0, s, CMat(A), (const int*)w.jpvt, w.jpvt_stride, d, R, n);
        DQ_HIP(hipGetLastError());
        return 0;
    }
    const size_t lds = sizeof(double) * ((size_t)3 * n + (size_t)16 * n + 32) + sizeof(int) * (16 + (size_t)n) + 64;
    hipLaunchKernelGGL((qrcp_kernel<NR>), dim3(1, n_chains), dim3(1024), lds, s, A, w, n);
    hipLaunchKernelGGL((formq_kernel<4 * NR>), dim3((n + 15) / 16, n_chains), dim3(256), 0, s, CMat(A), (const double*)w.tau, w.tau_stride,
                       (const int*)w.jpvt, w.jpvt_stride, L, n);
    hipLaunchKernelGGL(assemble_r_kernel, dim3(n, n_chains), dim3(128), 0, s, CMat(A), (const int*)w.jpvt, w.jpvt_stride, d, R, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

int qr_init_device() {
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qrcp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qrcp_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qrcp_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qrcp_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qrcp_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

int launch_to_ldr(Mat A, Mat L, Vec d, Mat R, QrWork w, int n, int n_chains, hipStream_t s) {
    if (n > 1024) { set_error("to_LDR kernel supports n <= 1024"); return -1; }
    static const bool force_stream = getenv("DQMC_QR_STREAMING") != nullptr;    // A/B switch for tests and profiling
    // panel-pivoted blocked QR (qr_panel.hip): one global pivot decision per 16 columns instead of one per column.  DQMC_QR_PANEL=0 keeps the
    // column-pivoted kernels at every size (test switch: the only way to reach qr_colown / qr_coop at n = 64 .. 1024 with few chains)
    static const bool use_panel = !(getenv("DQMC_QR_PANEL") && atoi(getenv("DQMC_QR_PANEL")) == 0);
    if (use_panel && !force_stream && n >= 64 && n_chains <= 8 && qr_panel_ok(n, w)) return launch_to_ldr_nr<-2>(A, L, d, R, w, n, n_chains, s);
    if (n <= 256 && !force_stream) return launch_to_ldr_nr<0>(A, L, d, R, w, n, n_chains, s);
    // n > 256: the matrix does not fit one CU; ceil(n/32) cooperating workgroups while they fit the CU budget (co-residency), the
    // single-workgroup streaming kernel otherwise (many chains per launch: every CU is busy with its own chain anyway)
    if (n > 256 && !force_stream && w.sync && w.sync_stride >= qrcp_coop_sync_granules(n) && qrcp_coop_workgroups(n, n_chains) <= 200)
        return launch_to_ldr_nr<-1>(A, L, d, R, w, n, n_chains, s);
    if (n <= 64) return launch_to_ldr_nr<1>(A, L, d, R, w, n, n_chains, s);
    if (n <= 128) return launch_to_ldr_nr<2>(A, L, d, R, w, n, n_chains, s);
    if (n <= 256) return launch_to_ldr_nr<4>(A, L, d, R, w, n, n_chains, s);
    if (n <= 576) return launch_to_ldr_nr<9>(A, L, d, R, w, n, n_chains, s);
    return launch_to_ldr_nr<16>(A, L, d, R, w, n, n_chains, s);
}

}  // namespace dq
