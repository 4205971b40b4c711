// elementwise.hip -- small batched memory-bound helpers around the GEMM chain.
//
// These carry the O(N^2) passes of the reference that are not fused into a
// GEMM epilogue: expV tables (source/model.cpp:62-84), the D_large/D_small
// split (source/stablelinalg.cpp:100-108), identity, copies, transposes, the
// TermA + TermB sum (:151) and the wrap-vs-stabilised max|dG|
// (DQMC::check_error, source/dqmc.cpp:317-329).  blockIdx.y = chain.
#include "common.h"
#include "wave.h"

namespace dq {

__global__ void build_expv_kernel(const int8_t* fields, long f_stride, long total, const double* tab,
                                  double* expv, double* invexpv, long v_stride) {
    const int c = blockIdx.y;
    const double* t = tab + (long)c * 8;
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
        const int f = fields[(long)c * f_stride + k] & 3;
        expv[(long)c * v_stride + k] = t[f];
        invexpv[(long)c * v_stride + k] = t[4 + f];
    }
}
int launch_build_expv(const int8_t* fields, long f_stride, int nt, int n, const double* tab, double* expv, double* invexpv,
                      long v_stride, int n_chains, hipStream_t s) {
    const long total = (long)nt * n;
    dim3 grid((unsigned)((total + 255) / 256 > 512 ? 512 : (total + 255) / 256), n_chains);
    hipLaunchKernelGGL(build_expv_kernel, grid, dim3(256), 0, s, fields, f_stride, total, tab, expv, invexpv, v_stride);
    DQ_HIP(hipGetLastError());
    return 0;
}

// one block of 256 threads per chain
__global__ void split_d_kernel(CVec d, Vec dl_inv, Vec ds, double* logsum, int n) {
    const int c = blockIdx.y;
    const double* dd = d.at(c);
    double ls = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = dd[i];
        const bool large = v >= 1.0;
        dl_inv.at(c)[i] = large ? 1.0 / v : 1.0;
        ds.at(c)[i] = large ? 1.0 : v;
        if (large) ls += log(v);
    }
    if (logsum) {
        __shared__ double red[256];
        red[threadIdx.x] = ls;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) logsum[c] = red[0];
    }
}
// two D splits in one launch (blockIdx.x picks the vector): inv_I_plus_ldr_mul_ldr needs both of its factors' splits, and a launch is
// ~4.5 us whatever it does
__global__ void split_d2_kernel(CVec d1, Vec dl1, Vec ds1, CVec d2, Vec dl2, Vec ds2, int n) {
    const int c = blockIdx.y;
    const double* dd = blockIdx.x ? d2.at(c) : d1.at(c);
    double* dl = blockIdx.x ? dl2.at(c) : dl1.at(c);
    double* ds = blockIdx.x ? ds2.at(c) : ds1.at(c);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = dd[i];
        const bool large = v >= 1.0;
        dl[i] = large ? 1.0 / v : 1.0;
        ds[i] = large ? 1.0 : v;
    }
}
int launch_split_d2(CVec d1, Vec dl1, Vec ds1, CVec d2, Vec dl2, Vec ds2, int n, int n_chains, hipStream_t s) {
    hipLaunchKernelGGL(split_d2_kernel, dim3(2, n_chains), dim3(256), 0, s, d1, dl1, ds1, d2, dl2, ds2, n);
    DQ_HIP(hipGetLastError());
    return 0;
}
int launch_split_d(CVec d, Vec dl_inv, Vec ds, double* logsum, int n, int n_chains, hipStream_t s) {
    hipLaunchKernelGGL(split_d_kernel, dim3(1, n_chains), dim3(256), 0, s, d, dl_inv, ds, logsum, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

__global__ void copy_kernel(CMat in, Mat out, long count) {
    const int c = blockIdx.y;
    const double* a = in.at(c); double* b = out.at(c);
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < count; k += (long)gridDim.x * blockDim.x) b[k] = a[k];
}
int launch_copy(CMat in, Mat out, long count, int n_chains, hipStream_t s) {
    long blocks = (count + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks, n_chains), dim3(256), 0, s, in, out, count);
    DQ_HIP(hipGetLastError());
    return 0;
}

__global__ void identity_kernel(Mat out, int n) {
    const int c = blockIdx.y; double* o = out.at(c);
    const long total = (long)n * n;
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x)
        o[k] = (k % n == k / n) ? 1.0 : 0.0;
}
int launch_set_identity(Mat out, int n, int n_chains, hipStream_t s) {
    long blocks = ((long)n * n + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(identity_kernel, dim3((unsigned)blocks, n_chains), dim3(256), 0, s, out, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

// 32x32 LDS-tiled transpose with optional row scale on the OUTPUT rows
__global__ void transpose_scale_kernel(CMat in, Mat out, CVec rs, int n, int tiles) {
    __shared__ double tile[32][33];
    const int c = blockIdx.y;
    const double* a = in.at(c); double* o = out.at(c);
    const int ti = blockIdx.x % tiles, tj = blockIdx.x / tiles;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = ti * 32 + tx, j = tj * 32 + r;
        tile[r][tx] = (i < n && j < n) ? a[i + (long)n * j] : 0.0;    // tile[jj][ii] = in[i][j]
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = tj * 32 + tx, j = ti * 32 + r;                  // out[i][j] = in[j][i]
        if (i < n && j < n) {
            double v = tile[tx][r];
            if (rs.p) v *= rs.at(c)[i];
            o[i + (long)n * j] = v;
        }
    }
}
int launch_transpose_scale(CMat in, Mat out, CVec rs, int n, int n_chains, hipStream_t s) {
    const int tiles = (n + 31) / 32;
    hipLaunchKernelGGL(transpose_scale_kernel, dim3(tiles * tiles, n_chains), dim3(256), 0, s, in, out, rs, n, tiles);
    DQ_HIP(hipGetLastError());
    return 0;
}

// max |A - B| per chain (NaN propagates).  Sixteen workgroups per chain stream the two matrices (one workgroup is bound by what a single
// CU pulls through its L1: 22 us for 1 MiB); wave maxima by DPP, block maxima folded with a 64-bit integer atomic max -- for
// non-negative doubles (and NaN) the IEEE bit pattern orders like the value, so the result does not depend on arrival order.
// The caller zeroes the err[] slots (one stream-ordered memset per half sweep) before the launches that target them.
__global__ __launch_bounds__(256) void max_abs_diff_kernel(CMat A, CMat B, double* err, long err_stride, int n) {
    const int c = blockIdx.y;
    const double* __restrict__ a = A.at(c); const double* __restrict__ b = B.at(c);
    const long total = (long)n * n;
    unsigned long long m = 0ULL;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
        const unsigned long long d = (unsigned long long)__double_as_longlong(fabs(a[k] - b[k]));      // |x| clears the sign bit: NaN stays the largest
        m = d > m ? d : m;
    }
    m = wave_max_u64(m);
    __shared__ unsigned long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long r = red[0];
        for (int q = 1; q < 4; ++q) r = red[q] > r ? red[q] : r;
        atomicMax(reinterpret_cast<unsigned long long*>(err + (long)c * err_stride), r);
    }
}
int launch_max_abs_diff(CMat A, CMat B, double* err, long err_stride, int n, int n_chains, hipStream_t s) {
    const long total = (long)n * n;
    const long want = (total + 4095) / 4096;                                  // 16 elements per thread
    const int blocks = (int)(want < 1 ? 1 : (want > 16 ? 16 : want));
    hipLaunchKernelGGL(max_abs_diff_kernel, dim3(blocks, n_chains), dim3(256), 0, s, A, B, err, err_stride, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

__global__ void add_scaled_cols_kernel(CMat A, CMat B, CVec cs, Mat C, int n) {
    const int c = blockIdx.y;
    const double* a = A.at(c); const double* b = B.at(c); double* o = C.at(c);
    const long total = (long)n * n;
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
        double bv = b[k];
        if (cs.p) bv *= cs.at(c)[k / n];
        o[k] = a[k] + bv;
    }
}
int launch_add_scaled_cols(CMat A, CMat B, CVec cs, Mat C, int n, int n_chains, hipStream_t s) {
    const long total = (long)n * n;
    long blocks = (total + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(add_scaled_cols_kernel, dim3((unsigned)blocks, n_chains), dim3(256), 0, s, A, B, cs, C, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

__global__ void scale_rows_kernel(CMat in, CVec rs, Mat out, int n) {
    const int c = blockIdx.y;
    const double* a = in.at(c); double* o = out.at(c); const double* r = rs.at(c);
    const long total = (long)n * n;
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) o[k] = a[k] * r[k % n];
}
int launch_scale_rows(CMat in, CVec rs, Mat out, int n, int n_chains, hipStream_t s) {
    const long total = (long)n * n;
    long blocks = (total + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks, n_chains), dim3(256), 0, s, in, rs, out, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

// DQMC::acc_rate_ += acc_l / nt per slice (source/dqmc.cpp:366,424); check_error's running
// max / sum / count (source/dqmc.cpp:324-326).  One thread per chain: the sums are tiny.
__global__ void fold_stats_kernel(DevStats* st, const int* acc, long acc_stride, int n_slices, const double* err, long err_stride,
                                  int n_err, int n, int nt, int n_chains) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    DevStats s = st[c];
    for (int k = 0; k < n_slices; ++k) {
        const int a = acc[(long)c * acc_stride + k];
        s.acc_rate += (static_cast<double>(a) / n) / nt;
        s.n_accepted += a; s.n_proposed += n;
    }
    for (int k = 0; k < n_err; ++k) {
        const double e = err[(long)c * err_stride + k];
        if (e > s.max_err || e != e) s.max_err = e;
        s.sum_err += e; s.n_err += 1.0;
    }
    st[c] = s;
}
int launch_fold_stats(DevStats* st, const int* acc, long acc_stride, int n_slices, const double* err, long err_stride, int n_err,
                      int n, int nt, int n_chains, hipStream_t s) {
    hipLaunchKernelGGL(fold_stats_kernel, dim3((n_chains + 63) / 64), dim3(64), 0, s, st, acc, acc_stride, n_slices, err, err_stride, n_err, n, nt, n_chains);
    DQ_HIP(hipGetLastError());
    return 0;
}


// ---- equal-time observables -------------------------------------------------------------------
// One workgroup per chain, thread t <-> site t AND displacement bin t (n_orb = 1: L1*L2 bins = n sites).
//   density   = (1/N) sum_i 2 (1 - G_ii)                                  source/model.cpp:167-192
//   doubleOcc = (1/N) sum_i (1 - G_ii)^2                                  :195-220
//   swave     = (1/N) sum_ij (delta_ji - G_ji)^2                          :222-256
//   ninj(i,j) = n_i n_j + 2 (1 - G_ji) G_ij - n_avg^2, n_i = 2 (1 - G_ii) :258-288 (the "1 -" is literal in the reference, also for i != j)
//   chi_r(d)  = (1/N) sum_i ninj(i, i + d)                                include/measurementh5.h:20-66
// Every bin is summed by one thread in a fixed order (no atomics): results are reproducible run to run.
__global__ __launch_bounds__(1024) void measure_equal_time_kernel(CMat Gm, double* out_p, long out_stride, int L1, int L2, int accumulate) {
    __shared__ double red[3][16];
    __shared__ double s_navg;
    const int chain = blockIdx.y;
    const int n = L1 * L2;
    const double* __restrict__ G = Gm.at(chain);
    double* out = out_p + (long)chain * out_stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = (blockDim.x + 63) >> 6;
    double dens = 0.0, docc = 0.0, sw = 0.0;
    if (t < n) {
        const double c = 1.0 - G[t + (long)n * t];
        dens = 2.0 * c; docc = c * c;
        for (int j = 0; j < n; ++j) { const double x = (j == t ? 1.0 : 0.0) - G[j + (long)n * t]; sw += x * x; }   // column t: contiguous
    }
    dens = wave_sum(dens); docc = wave_sum(docc); sw = wave_sum(sw);
    if (lane == 0) { red[0][wave] = dens; red[1][wave] = docc; red[2][wave] = sw; }
    __syncthreads();
    if (t == 0) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int q = 0; q < nwaves; ++q) { a += red[0][q]; b += red[1][q]; c += red[2][q]; }
        a /= n; b /= n; c /= n;
        s_navg = a;                                            // n_avg of calculate_densityCorr == density
        if (accumulate) { out[0] += a; out[1] += b; out[2] += c; } else { out[0] = a; out[1] = b; out[2] = c; }
    }
    __syncthreads();
    if (t < n) {
        // bin t = (dx_idx, dy_idx) = (t % L1, t / L1); displacement d = idx - (L/2 - 1) in (-L/2, L/2]
        const int dx = (t % L1) - (L1 / 2 - 1), dy = (t / L1) - (L2 / 2 - 1);
        const double navg = s_navg;
        double acc = 0.0;
        for (int i = 0; i < n; ++i) {
            const int xi = i % L1, yi = i / L1;
            const int xj = ((xi + dx) % L1 + L1) % L1, yj = ((yi + dy) % L2 + L2) % L2;
            const int j = yj * L1 + xj;
            const double ni = 2.0 * (1.0 - G[i + (long)n * i]), nj = 2.0 * (1.0 - G[j + (long)n * j]);
            acc += ni * nj + 2.0 * (1.0 - G[j + (long)n * i]) * G[i + (long)n * j] - navg * navg;
        }
        acc /= n;
        if (accumulate) out[3 + t] += acc; else out[3 + t] = acc;
    }
}

int launch_measure_equal_time(CMat G, double* out, long out_stride, int L1, int L2, int accumulate, int n_chains, hipStream_t s) {
    const int n = L1 * L2;
    if (n < 1 || n > 1024) { set_error("measure_equal_time: L1*L2 must be in 1..1024"); return -1; }
    const int threads = ((n + 63) / 64) * 64;
    hipLaunchKernelGGL(measure_equal_time_kernel, dim3(1, n_chains), dim3(threads), 0, s, G, out, out_stride, L1, L2, accumulate);
    DQ_HIP(hipGetLastError());
    return 0;
}


// out = a * in + b * I: G(0,tau=0) = G - I, G(beta,0) = I - G, G(0,beta) = -G of the unequal-time path (source/dqmc.cpp:236-239, 271-275)
__global__ void axpb_identity_kernel(CMat in, Mat out, double a, double b, int n) {
    const int chain = blockIdx.y;
    const double* x = in.at(chain);          // may alias out (in-place negation)
    double* y = out.at(chain);
    const long nn = (long)n * n;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nn; k += (long)gridDim.x * blockDim.x)
        y[k] = a * x[k] + ((k % n) == (k / n) ? b : 0.0);
}
int launch_axpb_identity(CMat in, Mat out, double a, double b, int n, int n_chains, hipStream_t s) {
    hipLaunchKernelGGL(axpb_identity_kernel, dim3(64, n_chains), dim3(256), 0, s, in, out, a, b, n);
    DQ_HIP(hipGetLastError());
    return 0;
}


// ---- dynamical observables -----------------------------------------------------------------------
// grid = (nt + 1 slices, chains), thread t <-> displacement bin t; per slice tau, with G00 = Gtt[0], spin up == spin down:
//   greenTau(i,j)   = 2 Gt0(i,j)                                              source/model.cpp:290-314
//   doublonTau(i,j) = Gt0(i,j)^2                                              :316-345
//   currxxTau(i,j)  = -(t1 - t2 - t3 + t4), ix = i + x, jx = j + x            :347-394
//       t1 = [2 Gtt(ix,i)][2 G00(jx,j)] - 2 G0t(jx,i) Gt0(ix,j)     t2 = [2 Gtt(ix,i)][2 G00(j,jx)] - 2 G0t(j,i) Gt0(ix,jx)
//       t3 = [2 Gtt(i,ix)][2 G00(jx,j)] - 2 G0t(jx,ix) Gt0(i,j)     t4 = [2 Gtt(i,ix)][2 G00(j,jx)] - 2 G0t(j,ix) Gt0(i,jx)
//   chi_r(d, tau) = (1/N) sum_i chi(i, i + d, tau)                            include/measurementh5.h:20-66
__global__ __launch_bounds__(1024) void measure_unequal_time_kernel(const double* Gtt_p, const double* Gt0_p, const double* G0t_p, double* out_p,
                                                                    long out_stride, int L1, int L2, int nt, int accumulate, int n_chains) {
    const int tau = blockIdx.x, chain = blockIdx.y;
    const int n = L1 * L2;
    const long nn = (long)n * n;
    const double* __restrict__ Gtt = Gtt_p + ((long)tau * n_chains + chain) * nn;
    const double* __restrict__ Gt0 = Gt0_p + ((long)tau * n_chains + chain) * nn;
    const double* __restrict__ G0t = G0t_p + ((long)tau * n_chains + chain) * nn;
    const double* __restrict__ G00 = Gtt_p + (long)chain * nn;
    double* out = out_p + (long)chain * out_stride;
    const int t = threadIdx.x;
    if (t >= n) return;
    const int dx = (t % L1) - (L1 / 2 - 1), dy = (t / L1) - (L2 / 2 - 1);
    double sg = 0.0, sd = 0.0, sc = 0.0;
#define AT(M, r, c) M[(r) + (long)n * (c)]
    for (int i = 0; i < n; ++i) {
        const int xi = i % L1, yi = i / L1;
        const int xj = ((xi + dx) % L1 + L1) % L1, yj = ((yi + dy) % L2 + L2) % L2;
        const int j = yj * L1 + xj;
        const int ix = yi * L1 + (xi + 1) % L1, jx = yj * L1 + (xj + 1) % L1;          // Lattice::site_neighbors(., {1,0}, 0), include/lattice.h:100-107
        const double g = AT(Gt0, i, j);
        sg += g + g; sd += g * g;
        const double dc1i = 2.0 * AT(Gtt, ix, i), dc2i = 2.0 * AT(Gtt, i, ix);
        const double dc1j = 2.0 * AT(G00, jx, j), dc2j = 2.0 * AT(G00, j, jx);
        const double c1 = 2.0 * AT(G0t, jx, i) * AT(Gt0, ix, j), c2 = 2.0 * AT(G0t, j, i) * AT(Gt0, ix, jx);
        const double c3 = 2.0 * AT(G0t, jx, ix) * g, c4 = 2.0 * AT(G0t, j, ix) * AT(Gt0, i, jx);
        sc += -((dc1i * dc1j - c1) - (dc1i * dc2j - c2) - (dc2i * dc1j - c3) + (dc2i * dc2j - c4));
    }
#undef AT
    const long slab = (long)(nt + 1) * n;
    double* o = out + (long)tau * n + t;
    if (accumulate) { o[0] += sg / n; o[slab] += sd / n; o[2 * slab] += sc / n; }
    else { o[0] = sg / n; o[slab] = sd / n; o[2 * slab] = sc / n; }
}
int launch_measure_unequal_time(const double* Gtt, const double* Gt0, const double* G0t, double* out, long out_stride, int L1, int L2, int nt,
                                int accumulate, int n_chains, hipStream_t s) {
    const int n = L1 * L2;
    if (n < 1 || n > 1024) { set_error("measure_unequal_time: L1*L2 must be in 1..1024"); return -1; }
    const int threads = ((n + 63) / 64) * 64;
    hipLaunchKernelGGL(measure_unequal_time_kernel, dim3(nt + 1, n_chains), dim3(threads), 0, s, Gtt, Gt0, G0t, out, out_stride, L1, L2, nt, accumulate, n_chains);
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
