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

// dynamic LDS layout: UW[kd][n] (double2 {U, W}), diag[n], dlt[n], rbv[n], ur[n], tables[32] | site[n] (int) | newf[n] (bytes)
//
// Everything about a proposal that does not depend on G is evaluated for the whole slice up
// front (thread j prepares proposal position j: site, delta, ratio prefactor, new field value,
// uniform), so the serial loop's critical path per proposal is one LDS read of diag[site] and
// ~6 flops.  The G column / row an accepted flip needs are prefetched from L2 eight proposals
// ahead into registers (two alternating groups of 8), for every proposal whether or not it
// ends up accepted: G in HBM/L2 does not change inside a window, so the prefetch is exact.
constexpr int SCAN_PF = 8;

struct ScanShared {
    double2* UW; double* diag; double* dlt; double* rbv; double* ur; double* tl; int* site; signed char* newf;
};

// loads G[:, site] / G[site, :] elements of thread j for the SCAN_PF proposals of group g
__device__ __forceinline__ void scan_prefetch(double (&pc)[SCAN_PF], double (&pr)[SCAN_PF], const double* __restrict__ G, const ScanShared& sh,
                                              int g, int n, int j, bool live) {
#pragma unroll
    for (int q = 0; q < SCAN_PF; ++q) {
        // unconditional loads from clamped addresses: a branch around a load makes hipcc wait for it at
        // the join (vmcnt(0) per element), which serialises the whole prefetch
        const int pos = min(g * SCAN_PF + q, n - 1);
        const int jj = live ? j : n - 1;
        const int i = sh.site[pos];
        pc[q] = G[jj + (long)n * i]; pr[q] = G[i + (long)n * jj];
    }
}

// processes the proposals of group g from position `pos` on; returns false when the window is full or the walk is over
#ifdef DQ_SCAN_STAMPS
#define STAMP(v) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); v = _t; }
struct ScanProf { unsigned long long t_rej = 0, t_kloop = 0, t_tail = 0, t_wait = 0; int n_rej = 0, n_acc = 0; };
#define PROF_ARG , ScanProf& prof
#else
#define PROF_ARG
#endif
__device__ __forceinline__ bool scan_group(const double (&pc)[SCAN_PF], const double (&pr)[SCAN_PF], const ScanShared& sh, int g, int n, int kd,
                                           int j, bool live, int& pos, int& k, const UpdateDesc& d, long slice_off, int chain, int8_t* fields_g,
                                           double* __restrict__ Up, double* __restrict__ Wp PROF_ARG) {
#pragma unroll
    for (int q = 0; q < SCAN_PF; ++q) {
        const int mypos = g * SCAN_PF + q;
        if (mypos < pos) continue;                        // resuming in the middle of a group
        if (mypos >= n || k >= kd) return false;
#ifdef DQ_SCAN_STAMPS
        unsigned long long t0, t1, t2, t3; STAMP(t0)
#endif
        const int i = sh.site[mypos];
        const double delta = sh.dlt[mypos];
        const double r = 1.0 + (1.0 - sh.diag[i]) * delta;        // det ratio per flavour (source/model.cpp:95)
        const double R = sh.rbv[mypos] * (r * r);                 // :121
        const double pacc = fmin(1.0, fabs(R));                   // source/update.cpp:24
        if (sh.ur[mypos] < pacc) {                                // bernoulli(p): u < p (include/utility.h:34-37)
            const double pref = delta / r;                        // source/model.cpp:132
            double uj = pc[q], wj = pr[q];
#ifdef DQ_SCAN_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(t1)
#endif
            if (live) {
                // G_eff[:, i] and G_eff[i, :]: one pass over the k <= 32 pending pairs.  A rolled loop is
                // bound by the LDS latency of each iteration (~100 ns x k); eight pairs per trip, all 16
                // reads in flight before the first use, four independent accumulator pairs.
                // 32-bit element offsets: LDS addresses are 32-bit, and size_t products per read cost more VALU
                // time than the reads themselves (measured: 2000 cycles per accepted flip before, see scripts/scan_stamps.py)
                const double2* own_p = sh.UW + j;                     // {U_m[j], W_m[j]} at own_p[m * n]
                const double2* piv_p = sh.UW + i;                     // {U_m[i], W_m[i]} (broadcast read)
                double u1 = 0.0, w1 = 0.0, u2 = 0.0, w2 = 0.0, u3 = 0.0, w3 = 0.0;
                int m = 0;
                for (; m + 8 <= k; m += 8) {
                    double2 o[8], pv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const int off = (m + e) * n; o[e] = own_p[off]; pv[e] = piv_p[off]; }
                    uj += o[0].x * pv[0].y; wj += pv[0].x * o[0].y; u1 += o[1].x * pv[1].y; w1 += pv[1].x * o[1].y;
                    u2 += o[2].x * pv[2].y; w2 += pv[2].x * o[2].y; u3 += o[3].x * pv[3].y; w3 += pv[3].x * o[3].y;
                    uj += o[4].x * pv[4].y; wj += pv[4].x * o[4].y; u1 += o[5].x * pv[5].y; w1 += pv[5].x * o[5].y;
                    u2 += o[6].x * pv[6].y; w2 += pv[6].x * o[6].y; u3 += o[7].x * pv[7].y; w3 += pv[7].x * o[7].y;
                }
                if (m < k) {                                          // tail of 1..7 pairs: load all, mask the excess
                    double2 o[8], pv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int mm = (m + e < k) ? m + e : m;      // clamp: in-range address, contribution masked below
                        const int off = mm * n; o[e] = own_p[off]; pv[e] = piv_p[off];
                    }
#pragma unroll
                    for (int e = 1; e < 8; ++e) if (m + e >= k) { o[e].x = 0.0; o[e].y = 0.0; }
                    uj += o[0].x * pv[0].y; wj += pv[0].x * o[0].y; u1 += o[1].x * pv[1].y; w1 += pv[1].x * o[1].y;
                    u2 += o[2].x * pv[2].y; w2 += pv[2].x * o[2].y; u3 += o[3].x * pv[3].y; w3 += pv[3].x * o[3].y;
                    uj += o[4].x * pv[4].y; wj += pv[4].x * o[4].y; u1 += o[5].x * pv[5].y; w1 += pv[5].x * o[5].y;
                    u2 += o[6].x * pv[6].y; w2 += pv[6].x * o[6].y; u3 += o[7].x * pv[7].y; w3 += pv[7].x * o[7].y;
                }
                uj += (u1 + u2) + u3; wj += (w1 + w2) + w3;
                if (j == i) wj -= 1.0;                            // V(i) -= 1 (source/model.cpp:135)
            }
#ifdef DQ_SCAN_STAMPS
            STAMP(t2)
#endif
            __syncthreads();                                      // every lane has read diag[i] and the pivot entries
            if (live) {
                const double pu = pref * uj;
                sh.UW[k * n + j] = make_double2(pu, wj);
                sh.diag[j] += pu * wj;
                Up[k * n + j] = pu; Wp[k * n + j] = wj;         // the flush kernel's operands, stored as we go (fire and forget)
                if (j == i) {
                    const int new_f = sh.newf[mypos];
                    fields_g[i] = (int8_t)new_f;                  // field.set_single_field (source/update.cpp:28)
                    d.expv[(long)chain * d.v_stride + slice_off + i] = sh.tl[24 + new_f];
                    d.invexpv[(long)chain * d.v_stride + slice_off + i] = sh.tl[28 + new_f];
                }
            }
            ++k;
            __syncthreads();
#ifdef DQ_SCAN_STAMPS
            STAMP(t3) prof.t_wait += t1 - t0; prof.t_kloop += t2 - t1; prof.t_tail += t3 - t2; prof.n_acc++;
        } else { STAMP(t1) prof.t_rej += t1 - t0; prof.n_rej++;
#endif
        }
        pos = mypos + 1;
    }
    return true;
}

// MAXT: the launch bound.  For n <= 256 the kernel runs one wave per SIMD and may use the whole 512-entry register
// file; under a 1024-thread bound (128 VGPRs) the prefetch groups spill around every accepted flip (measured:
// 2000 cycles per flip).
template <int MAXT>
__global__ __launch_bounds__(MAXT) void scan_kernel(UpdateDesc d, int l, int acc_slot, int first, int kd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int chain = blockIdx.y;
    const int n = d.n;
    const int j = threadIdx.x;
    const bool live = j < n;

    ScanShared sh;
    sh.UW = reinterpret_cast<double2*>(smem);
    sh.diag = reinterpret_cast<double*>(sh.UW + (size_t)kd * n);
    sh.dlt = sh.diag + n; sh.rbv = sh.dlt + n; sh.ur = sh.rbv + n; sh.tl = sh.ur + n;
    sh.site = reinterpret_cast<int*>(sh.tl + 32);
    sh.newf = reinterpret_cast<signed char*>(sh.site + n);

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
    if (live) {
        double* prep = d.prep + (long)chain * d.prep_stride;       // [4][n]: site | newf, dlt, rbv, ur of this slice
        if (first) {
            // proposal position j of this slice: site, old/new field value, ratio tables (source/model.cpp:99-122).
            // None of it depends on G or on earlier flips of the slice (each site is visited once).
            const long off = (long)chain * d.rs_stride + slice_off + j;
            const int i = d.perm[off];
            const int kp = d.kprop[off];
            const int old_f = fields_g[i];
            const int new_f = c_proposal[old_f][kp];
            const double rb = tab_g[old_f * 3 + kp], dl = tab_g[12 + old_f * 3 + kp], uu = d.u[off];
            sh.site[j] = i; sh.newf[j] = (signed char)new_f; sh.rbv[j] = rb; sh.dlt[j] = dl; sh.ur[j] = uu;
            prep[j] = __longlong_as_double(((long long)new_f << 32) | (unsigned int)i);
            prep[n + j] = dl; prep[2 * n + j] = rb; prep[3 * n + j] = uu;
        } else {
            const long long pk = __double_as_longlong(prep[j]);
            sh.site[j] = (int)(pk & 0xffffffffLL); sh.newf[j] = (signed char)(pk >> 32);
            sh.dlt[j] = prep[n + j]; sh.rbv[j] = prep[2 * n + j]; sh.ur[j] = prep[3 * n + j];
        }
        sh.diag[j] = G[j + (long)n * j];
    }
    if (j < 32) sh.tl[j] = tab_g[j];
    __syncthreads();

    int k = 0;
    double* __restrict__ Up = d.Upanel + (long)chain * d.panel_stride;
    double* __restrict__ Wp = d.Wpanel + (long)chain * d.panel_stride;
#ifdef DQ_SCAN_STAMPS
    ScanProf prof; unsigned long long tk0; STAMP(tk0)
#define PROF_PASS , prof
#else
#define PROF_PASS
#endif
    double pcA[SCAN_PF], prA[SCAN_PF], pcB[SCAN_PF], prB[SCAN_PF];
    int g = pos / SCAN_PF;
    scan_prefetch(pcA, prA, G, sh, g, n, j, live);
    scan_prefetch(pcB, prB, G, sh, g + 1, n, j, live);
    for (;;) {
        if (!scan_group(pcA, prA, sh, g, n, kd, j, live, pos, k, d, slice_off, chain, fields_g, Up, Wp PROF_PASS)) break;
        scan_prefetch(pcA, prA, G, sh, g + 2, n, j, live);
        if (!scan_group(pcB, prB, sh, g + 1, n, kd, j, live, pos, k, d, slice_off, chain, fields_g, Up, Wp PROF_PASS)) break;
        scan_prefetch(pcB, prB, G, sh, g + 3, n, j, live);
        g += 2;
    }
    if (j == 0) {
        state[0] = pos;
        state[1] = k;
        int* acc = d.acc_out + (long)chain * d.acc_stride + acc_slot;
        *acc = (first ? 0 : *acc) + k;
#ifdef DQ_SCAN_STAMPS
        unsigned long long tk1; STAMP(tk1)
        printf("scan l=%d first=%d: total %llu cyc | %d rej %llu cyc | %d acc: wait %llu kloop %llu tail %llu\n", l, first, tk1 - tk0,
               prof.n_rej, prof.t_rej, prof.n_acc, prof.t_wait, prof.t_kloop, prof.t_tail);
#endif
    }
}

// G[a,b] += sum_{m<k} Up[m][a] * Wp[m][b];  one wave per 16x16 tile, 2x2 waves per block.
// Latency matters more than flops here (N x k x N with k <= 32): the window's pair count k, the G tile and the
// panel rows are all loaded in ONE round trip (panel rows unconditionally for all KD slots -- stale slots are
// masked after the fact), instead of k -> panels -> G one after the other.
__global__ __launch_bounds__(256) void flush_kernel(UpdateDesc d, int tiles_per_dim, int kd) {
    const int chain = blockIdx.y;
    const int n = d.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a0 = (blockIdx.x % tiles_per_dim) * 32 + (wave & 1) * 16;
    const int b0 = (blockIdx.x / tiles_per_dim) * 32 + (wave >> 1) * 16;
    if (a0 >= n || b0 >= n) return;
    const double* __restrict__ Up = d.Upanel + (long)chain * d.panel_stride;
    const double* __restrict__ Wp = d.Wpanel + (long)chain * d.panel_stride;
    double* __restrict__ G = d.G.at(chain);
    const int r = lane & 15, kk = lane >> 4;
    const int a = min(a0 + r, n - 1), b = min(b0 + r, n - 1);          // clamped: loads stay unconditional
    const bool a_ok = a0 + r < n, b_ok = b0 + r < n;
    const int k = d.state[(long)chain * d.state_stride + 1];
    double uv[UPDATE_KD / 4], wv[UPDATE_KD / 4], gv[4];
#pragma unroll
    for (int s = 0; s < UPDATE_KD / 4; ++s) {
        const int m = min(4 * s + kk, kd - 1);
        wv[s] = Wp[m * n + b]; uv[s] = Up[m * n + a];
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) { const int bb = min(b0 + kk + 4 * reg, n - 1); gv[reg] = G[a + (long)n * bb]; }
    if (k == 0) return;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < UPDATE_KD / 4; ++s) {
        const int m = 4 * s + kk;
        const bool ok = m < k;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ok && b_ok) ? wv[s] : 0.0, (ok && a_ok) ? uv[s] : 0.0, acc, 0, 0, 0);   // A: rows <-> b, B: cols <-> a
    }
    if (a_ok) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int bb = b0 + kk + 4 * reg;
            if (bb < n) G[a + (long)n * bb] = gv[reg] + acc[reg];
        }
    }
}

static int pick_kd(int n) {
    // LDS budget: 16*kd*n + n*(4*8+4+1) + tables <= ~150 KiB
    const long budget = 150 * 1024 - (long)n * 40 - 512;
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
    const size_t lds = (size_t)16 * kd * n + (size_t)n * 32 + 256 + (size_t)n * 4 + (size_t)n + 64;
    static bool attr_set = false;
    if (!attr_set) {
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(scan_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(scan_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int windows = (n + kd - 1) / kd;
    const int tiles = (n + 31) / 32;
    for (int w = 0; w < windows; ++w) {
        if (threads <= 256) hipLaunchKernelGGL(scan_kernel<256>, dim3(1, n_chains), dim3(threads), lds, s, d, l, acc_slot, w == 0 ? 1 : 0, kd);
        else hipLaunchKernelGGL(scan_kernel<1024>, dim3(1, n_chains), dim3(threads), lds, s, d, l, acc_slot, w == 0 ? 1 : 0, kd);
        hipLaunchKernelGGL(flush_kernel, dim3(tiles * tiles, n_chains), dim3(256), 0, s, d, tiles, kd);
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
