// qr.hip -- stablelinalg::to_LDR (source/stablelinalg.cpp:35-55) on device:
// Householder QR with column pivoting (the dgeqp3 behind arma::qr(...,"vector")),
// explicit Q (dorgqr), d = |diag R0| and R = diag(1/d) R0 P^T.
//
//   qrcp_kernel   one 1024-thread workgroup per chain.  LAPACK dlaqp2
//                 semantics: pivot = remaining column of largest partial norm
//                 (lowest index on ties), Householder reflector from dlarfg
//                 (beta = -sign(alpha)*norm), partial norms down-dated and
//                 recomputed on cancellation.  The trailing matrix streams
//                 from L2; wave w owns trailing columns i+1+w, i+1+w+16, ...,
//                 a lane owns rows lane, lane+64, ... of a column, so global
//                 accesses are 512-byte coalesced and the per-column dot is a
//                 wave reduction.
//   formq_kernel  Q = H_0 ... H_{n-1} applied to I.  Column tiles of Q are
//                 independent, so the grid is (n/16 tiles) x chains and every
//                 wave keeps its 4 columns in registers: no LDS, no barriers.
//   assemble_r_kernel  d and the un-pivoted, row-normalised R.
#include "common.h"

namespace dq {

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// NR = rows per lane (n <= 64*NR)
template <int NR>
__global__ __launch_bounds__(1024) void qrcp_kernel(Mat Am, QrWork w, int n) {
    __shared__ double vn1[1024], vn2[1024], v[1024];
    __shared__ double red[16];
    __shared__ int s_pvt;
    __shared__ double s_alpha;
    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    double* tau = w.tau + (long)chain * w.tau_stride;
    int* jpvt = w.jpvt + (long)chain * w.jpvt_stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double tol3z = 1.0536712127723509e-08;    // sqrt(dlamch('Epsilon')) = sqrt(2^-53)

    // initial column norms
    for (int c = wave; c < n; c += 16) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NR; ++q) { const int r = lane + 64 * q; if (r < n) { const double x = A[r + (long)n * c]; s += x * x; } }
        s = wave_sum(s);
        if (lane == 0) { const double nn = sqrt(s); vn1[c] = nn; vn2[c] = nn; }
    }
    if (t < n) jpvt[t] = t;
    __syncthreads();

    for (int i = 0; i < n; ++i) {
        // (a) pivot: arg max of vn1[i..n), lowest index on ties
        if (wave == 0) {
            double best = -1.0; int bi = i;
            for (int c = i + lane; c < n; c += 64) { const double x = vn1[c]; if (x > best) { best = x; bi = c; } }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == 0) s_pvt = bi;
        }
        __syncthreads();
        const int pvt = s_pvt;
        // (b) swap columns i <-> pvt; thread t keeps the new A[t, i]
        double x = 0.0;
        if (t < n) {
            x = A[t + (long)n * pvt];
            if (pvt != i) { const double y = A[t + (long)n * i]; A[t + (long)n * pvt] = y; A[t + (long)n * i] = x; }
        }
        if (t == 0 && pvt != i) { const int jp = jpvt[pvt]; jpvt[pvt] = jpvt[i]; jpvt[i] = jp; vn1[pvt] = vn1[i]; vn2[pvt] = vn2[i]; }
        // (c) Householder vector (dlarfg)
        double ss = (t > i && t < n) ? x * x : 0.0;
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        if (t == i) s_alpha = x;
        __syncthreads();
        double xnorm2 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) xnorm2 += red[k];
        const double al = s_alpha;
        double tau_i = 0.0, beta = al, scale = 0.0;
        if (xnorm2 != 0.0) {
            beta = -copysign(sqrt(al * al + xnorm2), al);
            tau_i = (beta - al) / beta;
            scale = 1.0 / (al - beta);
        }
        if (t < n) {
            double vr = 0.0;
            if (t == i) { vr = 1.0; A[t + (long)n * i] = beta; }
            else if (t > i) { vr = x * scale; A[t + (long)n * i] = vr; }
            v[t] = vr;
        }
        if (t == 0) tau[i] = tau_i;
        __syncthreads();
        // (d) apply H to trailing columns, down-date norms
        double vr[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) { const int r = lane + 64 * q; vr[q] = (r < n) ? v[r] : 0.0; }
        for (int c = i + 1 + wave; c < n; c += 16) {
            double a[NR]; double s = 0.0;
#pragma unroll
            for (int q = 0; q < NR; ++q) { const int r = lane + 64 * q; a[q] = (r >= i && r < n) ? A[r + (long)n * c] : 0.0; s += a[q] * vr[q]; }
            s = wave_sum(s) * tau_i;
            double tail = 0.0, aic = 0.0;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int r = lane + 64 * q;
                if (r >= i && r < n) {
                    a[q] -= s * vr[q];
                    if (tau_i != 0.0) A[r + (long)n * c] = a[q];
                    if (r > i) tail += a[q] * a[q]; else aic = a[q];
                }
            }
            // norm down-date (dlaqp2)
            aic = wave_sum(aic);                 // exactly one lane holds row i
            const double n1 = vn1[c];
            if (n1 != 0.0) {
                double temp = fabs(aic) / n1; temp = fmax(0.0, 1.0 - temp * temp);
                const double rr = n1 / vn2[c];
                const double temp2 = temp * rr * rr;
                if (temp2 <= tol3z) {
                    tail = wave_sum(tail);
                    const double nn = (i + 1 < n) ? sqrt(tail) : 0.0;
                    if (lane == 0) { vn1[c] = nn; vn2[c] = nn; }
                } else if (lane == 0) vn1[c] = n1 * sqrt(temp);
            }
        }
        __syncthreads();
    }
}

// Each wave owns CW columns of Q in registers; applies H_i, i = n-1 .. 0.
template <int NR>
__global__ __launch_bounds__(256) void formq_kernel(CMat Am, const double* tau_p, long tau_stride, Mat Qm, int n) {
    constexpr int CW = 4;
    const int chain = blockIdx.y;
    const double* __restrict__ A = Am.at(chain);
    const double* __restrict__ tau = tau_p + (long)chain * tau_stride;
    double* __restrict__ Q = Qm.at(chain);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 16 + wave * CW;
    if (c0 >= n) return;
    double q[CW][NR];
#pragma unroll
    for (int cc = 0; cc < CW; ++cc)
#pragma unroll
        for (int k = 0; k < NR; ++k) q[cc][k] = (lane + 64 * k == c0 + cc) ? 1.0 : 0.0;
    const int cmax = min(c0 + CW - 1, n - 1);
    for (int i = cmax; i >= 0; --i) {
        const double ti = tau[i];
        if (ti == 0.0) continue;
        double vr[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) { const int r = lane + 64 * k; vr[k] = (r > i && r < n) ? A[r + (long)n * i] : (r == i ? 1.0 : 0.0); }
#pragma unroll
        for (int cc = 0; cc < CW; ++cc) {
            if (c0 + cc < i) continue;            // column still e_c: untouched by H_i
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NR; ++k) s += vr[k] * q[cc][k];
            s = wave_sum(s) * ti;
#pragma unroll
            for (int k = 0; k < NR; ++k) q[cc][k] -= s * vr[k];
        }
    }
#pragma unroll
    for (int cc = 0; cc < CW; ++cc) {
        if (c0 + cc >= n) continue;
#pragma unroll
        for (int k = 0; k < NR; ++k) { const int r = lane + 64 * k; if (r < n) Q[r + (long)n * (c0 + cc)] = q[cc][k]; }
    }
}

// d[j] = |R0[j,j]|;  R[:, jpvt[j]] = R0[:, j] / d   (source/stablelinalg.cpp:47-52)
__global__ void assemble_r_kernel(CMat Am, const int* jpvt_p, long jpvt_stride, Vec dv, Mat Rm, int n) {
    const int chain = blockIdx.y;
    const double* __restrict__ A = Am.at(chain);
    const int* jpvt = jpvt_p + (long)chain * jpvt_stride;
    double* __restrict__ R = Rm.at(chain);
    double* d = dv.at(chain);
    const int j = blockIdx.x;                     // source column
    const int dst = jpvt[j];
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        double val = 0.0;
        if (r <= j) val = A[r + (long)n * j] / fabs(A[r + (long)n * r]);
        R[r + (long)n * dst] = val;
    }
    if (threadIdx.x == 0) d[j] = fabs(A[j + (long)n * j]);
}

template <int NR>
static int launch_to_ldr_nr(Mat A, Mat L, Vec d, Mat R, QrWork w, int n, int n_chains, hipStream_t s) {
    hipLaunchKernelGGL((qrcp_kernel<NR>), dim3(1, n_chains), dim3(1024), 0, s, A, w, n);
    hipLaunchKernelGGL((formq_kernel<NR>), dim3((n + 15) / 16, n_chains), dim3(256), 0, s, CMat(A), (const double*)w.tau, w.tau_stride, L, n);
    hipLaunchKernelGGL(assemble_r_kernel, dim3(n, n_chains), dim3(128), 0, s, CMat(A), (const int*)w.jpvt, w.jpvt_stride, d, R, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

int launch_to_ldr(Mat A, Mat L, Vec d, Mat R, QrWork w, int n, int n_chains, hipStream_t s) {
    if (n > 1024) { set_error("to_LDR kernel supports n <= 1024"); return -1; }
    if (n <= 64) return launch_to_ldr_nr<1>(A, L, d, R, w, n, n_chains, s);
    if (n <= 128) return launch_to_ldr_nr<2>(A, L, d, R, w, n, n_chains, s);
    if (n <= 256) return launch_to_ldr_nr<4>(A, L, d, R, w, n, n_chains, s);
    if (n <= 576) return launch_to_ldr_nr<9>(A, L, d, R, w, n, n_chains, s);
    return launch_to_ldr_nr<16>(A, L, d, R, w, n, n_chains, s);
}

}  // namespace dq
