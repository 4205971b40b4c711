// update.hip -- update::local_update (source/update.cpp:5-32) for one time
// slice: the serial Metropolis walk over sites with the Sherman-Morrison
// Green's-function update (AttractiveHubbard::local_update_ratio /
// update_greens_local, source/model.cpp:90-138).
//
// Structure (one workgroup per Markov chain; blockIdx.y = chain, thread j <-> site j)
//   scan kernel : walks the pre-drawn site order.  The acceptance ratio needs
//                 only G_ii, which is kept in an LDS copy of the diagonal, so a
//                 rejected proposal costs one LDS read and a handful of fp64
//                 ops, evaluated redundantly by every lane (wave-uniform
//                 control flow, no broadcast).  An accepted flip is NOT
//                 applied to G in HBM; it is appended to an LDS-resident
//                 low-rank correction  G_eff = G + sum_m U_m W_m^T  (the
//                 "delayed update" the reference README lists as future work,
//                 README.md:41):  u = G_eff[:,i], w = G_eff[i,:] - e_i are
//                 formed with one dot-product sweep over the <= KD pairs, and
//                 the LDS diagonal is advanced by pref*u_j*w_j.  The Markov
//                 chain is the reference's; only the rounding order of the
//                 rank-1 sums differs.
//   flush kernel: whole chip, fp64 MFMA:  G += U^T-panel x W-panel  (N x k x N).
// A slice is ceil(N/KD) (scan, flush) pairs; pairs past the end of the walk
// find pos == N / k == 0 and exit immediately.
#include "common.h"

namespace dq {

using d4 = __attribute__((ext_vector_type(4))) double;

__constant__ int c_proposal[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};   // include/field.h:45-48

// dynamic LDS layout (doubles first, then ints, then bytes)
//   U[kd][n], W[kd][n], diag[n], u_rand[n], tables[32] | perm[n] (int) | kprop[n], f[n] (bytes)
__global__ __launch_bounds__(1024) void scan_kernel(UpdateDesc d, int l, int acc_slot, int first, int kd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int chain = blockIdx.y;
    const int n = d.n;
    const int j = threadIdx.x;
    const bool live = j < n;

    double* U = reinterpret_cast<double*>(smem);
    double* W = U + (size_t)kd * n;
    double* diag = W + (size_t)kd * n;
    double* urand = diag + n;
    double* tl = urand + n;                       // 32 doubles: rb[12], delta[12], ev[4], iev[4]
    int* perm = reinterpret_cast<int*>(tl + 32);
    unsigned char* kprop = reinterpret_cast<unsigned char*>(perm + n);
    signed char* f = reinterpret_cast<signed char*>(kprop + n);

    double* __restrict__ G = d.G.at(chain);
    int* state = d.state + (long)chain * d.state_stride;
    const double* tab_g = reinterpret_cast<const double*>(d.tabs + chain);
    const long slice_off = (long)l * n;
    int8_t* fields_g = d.fields + (long)chain * d.f_stride + slice_off;

    int pos = first ? 0 : state[0];
    if (pos >= n) {                     // walk already finished: nothing to flush
        if (j == 0) state[1] = 0;
        return;
    }
    if (j < 32) tl[j] = tab_g[j];
    if (live) {
        diag[j] = G[j + (long)n * j];
        urand[j] = d.u[(long)chain * d.rs_stride + slice_off + j];
        perm[j] = d.perm[(long)chain * d.rs_stride + slice_off + j];
        kprop[j] = d.kprop[(long)chain * d.rs_stride + slice_off + j];
        f[j] = fields_g[j];
    }
    __syncthreads();

    int k = 0;
    while (pos < n && k < kd) {
        const int i = perm[pos];
        const int old_f = f[i];
        const int kp = kprop[pos];
        const double rb = tl[old_f * 3 + kp];
        const double delta = tl[12 + old_f * 3 + kp];
        const double r = 1.0 + (1.0 - diag[i]) * delta;          // det ratio per flavour (source/model.cpp:95)
        const double R = rb * (r * r);                            // :121
        const double p = fmin(1.0, fabs(R));                      // source/update.cpp:24
        if (urand[pos] < p) {                                     // bernoulli(p): u < p (include/utility.h:34-37)
            const double pref = delta / r;                        // source/model.cpp:132
            double uj = 0.0, wj = 0.0;
            if (live) {
                uj = G[j + (long)n * i];
                wj = G[i + (long)n * j];
                for (int m = 0; m < k; ++m) {
                    uj += U[(size_t)m * n + j] * W[(size_t)m * n + i];
                    wj += U[(size_t)m * n + i] * W[(size_t)m * n + j];
                }
                if (j == i) wj -= 1.0;                            // V(i) -= 1 (source/model.cpp:135)
            }
            __syncthreads();                                      // all reads of diag[i], f[i] done
            if (live) {
                const double pu = pref * uj;
                U[(size_t)k * n + j] = pu;
                W[(size_t)k * n + j] = wj;
                diag[j] += pu * wj;
                if (j == i) {
                    const int new_f = c_proposal[old_f][kp];
                    f[i] = (signed char)new_f;
                    fields_g[i] = (int8_t)new_f;                  // field.set_single_field (source/update.cpp:28)
                    d.expv[(long)chain * d.v_stride + slice_off + i] = tl[24 + new_f];
                    d.invexpv[(long)chain * d.v_stride + slice_off + i] = tl[28 + new_f];
                }
            }
            ++k;
            __syncthreads();
        }
        ++pos;
    }
    // hand the window's panels to the flush kernel
    if (live) {
        double* Up = d.Upanel + (long)chain * d.panel_stride;
        double* Wp = d.Wpanel + (long)chain * d.panel_stride;
        for (int m = 0; m < k; ++m) { Up[(size_t)m * n + j] = U[(size_t)m * n + j]; Wp[(size_t)m * n + j] = W[(size_t)m * n + j]; }
    }
    if (j == 0) {
        state[0] = pos;
        state[1] = k;
        int* acc = d.acc_out + (long)chain * d.acc_stride + acc_slot;
        *acc = (first ? 0 : *acc) + k;
    }
}

// G[a,b] += sum_{m<k} Up[m][a] * Wp[m][b];  one wave per 16x16 tile, 2x2 waves per block.
__global__ __launch_bounds__(256) void flush_kernel(UpdateDesc d, int tiles_per_dim) {
    const int chain = blockIdx.y;
    const int k = d.state[(long)chain * d.state_stride + 1];
    if (k == 0) return;
    const int n = d.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a0 = (blockIdx.x % tiles_per_dim) * 32 + (wave & 1) * 16;
    const int b0 = (blockIdx.x / tiles_per_dim) * 32 + (wave >> 1) * 16;
    if (a0 >= n || b0 >= n) return;
    const double* __restrict__ Up = d.Upanel + (long)chain * d.panel_stride;
    const double* __restrict__ Wp = d.Wpanel + (long)chain * d.panel_stride;
    double* __restrict__ G = d.G.at(chain);
    const int r = lane & 15, kk = lane >> 4;
    const int a = a0 + r, b = b0 + r;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int m0 = 0; m0 < k; m0 += 4) {
        const int m = m0 + kk;
        const double wv = (m < k && b < n) ? Wp[(size_t)m * n + b] : 0.0;    // MFMA-A: rows <-> b
        const double uv = (m < k && a < n) ? Up[(size_t)m * n + a] : 0.0;    // MFMA-B: cols <-> a
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, uv, acc, 0, 0, 0);
    }
    if (a < n) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int bb = b0 + kk + 4 * reg;
            if (bb < n) G[a + (long)n * bb] += acc[reg];
        }
    }
}

static int pick_kd(int n) {
    // LDS budget: 2*kd*n*8 + n*(8+8+4+1+1) <= ~150 KiB
    const long budget = 150 * 1024 - (long)n * 24 - 512;
    long kd = budget / (16L * n);
    if (kd > UPDATE_KD) kd = UPDATE_KD;
    if (kd < 1) kd = 1;
    return (int)kd;
}

int launch_update_slice(const UpdateDesc& d, int l, int acc_slot, int n_chains, hipStream_t s) {
    const int n = d.n;
    if (n > 1024) { set_error("local update kernel supports n_sites <= 1024"); return -1; }
    const int kd = pick_kd(n);
    const int threads = ((n + 63) / 64) * 64;
    const size_t lds = (size_t)2 * kd * n * 8 + (size_t)n * 16 + 256 + (size_t)n * 4 + (size_t)n * 2 + 64;
    static bool attr_set = false;
    if (!attr_set) {
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int windows = (n + kd - 1) / kd;
    const int tiles = (n + 31) / 32;
    for (int w = 0; w < windows; ++w) {
        hipLaunchKernelGGL(scan_kernel, dim3(1, n_chains), dim3(threads), lds, s, d, l, acc_slot, w == 0 ? 1 : 0, kd);
        hipLaunchKernelGGL(flush_kernel, dim3(tiles * tiles, n_chains), dim3(256), 0, s, d, tiles);
    }
    DQ_HIP(hipGetLastError());
    return 0;
}

// ---- standalone rank-1 update (AttractiveHubbard::update_greens_local) -------
// scratch per chain: u[n], w[n], pref
__global__ void rank1_gather_kernel(Mat G, int i, double delta, double* scratch, long scratch_stride, int n) {
    const int c = blockIdx.y;
    const double* g = G.at(c);
    double* sc = scratch + (long)c * scratch_stride;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        sc[j] = g[j + (long)n * i];
        sc[n + j] = g[i + (long)n * j] - (j == i ? 1.0 : 0.0);
    }
    if (threadIdx.x == 0) sc[2 * n] = delta / (1.0 + (1.0 - g[i + (long)n * i]) * delta);
}
__global__ void rank1_apply_kernel(Mat G, const double* scratch, long scratch_stride, int n) {
    const int c = blockIdx.y;
    double* g = G.at(c);
    const double* sc = scratch + (long)c * scratch_stride;
    const double pref = sc[2 * n];
    const long total = (long)n * n;
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
        const int a = (int)(k % n), b = (int)(k / n);
        g[k] += pref * sc[a] * sc[n + b];
    }
}
int launch_rank1(Mat G, int i, double delta, double* scratch, long scratch_stride, int n, int n_chains, hipStream_t s) {
    hipLaunchKernelGGL(rank1_gather_kernel, dim3(1, n_chains), dim3(256), 0, s, G, i, delta, scratch, scratch_stride, n);
    long blocks = ((long)n * n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rank1_apply_kernel, dim3((unsigned)blocks, n_chains), dim3(256), 0, s, G, (const double*)scratch, scratch_stride, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
