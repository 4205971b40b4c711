// update.hip -- update::local_update (source/update.cpp:5-32) for one time
// slice: the serial Metropolis walk over sites with the Sherman-Morrison
// Green's-function update (AttractiveHubbard::local_update_ratio /
// update_greens_local, source/model.cpp:90-138).
//
// The acceptance ratio needs only G_ii, which is kept in an LDS copy of the diagonal.  An accepted flip is NOT applied
// to G in HBM; it is appended to a low-rank correction  G_eff = G + sum_m U_m W_m^T  (the "delayed update" the
// reference README lists as future work, README.md:41):  u = G_eff[:,i], w = G_eff[i,:] - e_i are formed with one pass
// over the <= KD pending pairs, and the diagonal is advanced by pref*u_j*w_j.  After KD = 32 flips (a "window") the
// correction is flushed into G with fp64 MFMAs:  G += U^T-panel x W-panel  (N x k x N).  The Markov chain is the
// reference's; only the rounding order of the rank-1 sums differs.
//
// Kernels (blockIdx.y = chain, thread j <-> site j):
//   slice_kernel        n <= 256, few chains: the whole slice in ONE launch -- workgroup 0 walks (walk_window6), the other
//                       (n/32)^2 workgroups each own a tile of G and of its transposed copy GT and flush every window;
//                       hand-offs through tagged words (SliceSync, common.h), see the comment at the kernel.
//   slice_solo_kernel   many chains: one workgroup per chain walks AND flushes.
//   scan_kernel<256> +  the same walk / flush code as separate launches, ceil(N/KD) pairs per slice (pairs past the end of
//   flush_kernel<true>  the walk find pos == N / k == 0 and exit); used when the single-launch grid would not be co-resident.
//   scan_kernel<1024> + n > 256: the pending pairs live in LDS (scan_group), G rows are read directly (no GT).
//   flush_kernel<false>
#include "common.h"
#include <cstdlib>
#include <mutex>
#include "wave.h"

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

// agent-scope relaxed accesses (sc1 on gfx950: bypass the CU's L1, write through): what the persistent
// slice kernel uses for every word another workgroup of the same launch writes or reads
__device__ __forceinline__ double ld_coh(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_coh(double* p, double x) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bytes of dynamic LDS: UW[kd][n] | diag, dlt, rbv, ur [n] | tables[32] | site[n] | newf[n] (padded) | (register variant) diag2[n]
__host__ __device__ inline size_t scan_lds_bytes(int n, int kd, bool regs) {
    const size_t base = (((size_t)16 * kd * n + (size_t)n * 32 + 256 + (size_t)n * 4 + (size_t)n) + 63) & ~(size_t)63;
    return base + (regs ? (size_t)n * 8 + 4 * (UPDATE_KD + 4) : 0);       // diag2 | acc_site[KD] | 4 spare words (verdicts of the persistent kernel's wave 0)
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release fence, which on
// gfx950 is s_waitcnt vmcnt(0): every barrier of the walk would wait for the panel stores just issued (~500-1000
// clk to L2 / memory) and for the G prefetch of the NEXT groups, i.e. the prefetch could never run ahead (measured:
// tail 550 -> 1100 clk per accepted flip once it became the only barrier).  Global data written in the walk is
// consumed by other kernels / other workgroups after an explicit vmcnt(0), never by this workgroup through memory.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct ScanShared {
    double2* UW; double* diag; double* dlt; double* rbv; double* ur; double* tl; int* site; signed char* newf;
    double* diag2 = nullptr;      // walk v3 only: second diagonal buffer
    int* acc_site = nullptr;      // walk v3 only: proposal index of pending pair m (KD entries)
};

// loads G[:, site] / G[site, :] elements of thread j for the SCAN_PF proposals of group g
template <bool COH>
__device__ __forceinline__ void scan_prefetch(double (&pc)[SCAN_PF], double (&pr)[SCAN_PF], const double* __restrict__ G, const ScanShared& sh,
                                              int g, int n, int j, bool live) {
#pragma unroll
    for (int q = 0; q < SCAN_PF; ++q) {
        // unconditional loads from clamped addresses: a branch around a load makes hipcc wait for it at
        // the join (vmcnt(0) per element), which serialises the whole prefetch
        const int pos = min(g * SCAN_PF + q, n - 1);
        const int jj = live ? j : n - 1;
        const int i = sh.site[pos];
        if (COH) { pc[q] = ld_coh(G + jj + (long)n * i); pr[q] = ld_coh(G + i + (long)n * jj); }
        else { pc[q] = G[jj + (long)n * i]; pr[q] = G[i + (long)n * jj]; }
    }
}

// processes the proposals of group g from position `pos` on; returns false when the window is full or the walk is over
#ifdef DQ_SCAN_STAMPS
#define STAMP(v) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); v = _t; }
struct ScanProf { unsigned long long t_rej = 0, t_kloop = 0, t_tail = 0, t_wait = 0, t_first = 0, t_dump = 0, t_noacc = 0, t_pub = 0, t_fin = 0, t_finwait = 0; int n_rej = 0, n_acc = 0, n_noacc = 0, n_pub = 0, n_fin_k = 0, n_fin_g = 0, n_fin_e = 0; };
#define PROF_ARG , ScanProf& prof
#else
#define PROF_ARG
#endif
template <bool COH>
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
            lds_barrier();                                      // every lane has read diag[i] and the pivot entries
            if (live) {
                const double pu = pref * uj;
                sh.UW[k * n + j] = make_double2(pu, wj);
                sh.diag[j] += pu * wj;
                if (COH) { st_coh(Up + k * n + j, pu); st_coh(Wp + k * n + j, wj); }
                else { Up[k * n + j] = pu; Wp[k * n + j] = wj; }    // the flush kernel's operands, stored as we go (fire and forget)
                if (j == i) {
                    const int new_f = sh.newf[mypos];
                    fields_g[i] = (int8_t)new_f;                  // field.set_single_field (source/update.cpp:28)
                    d.expv[(long)chain * d.v_stride + slice_off + i] = sh.tl[24 + new_f];
                    d.invexpv[(long)chain * d.v_stride + slice_off + i] = sh.tl[28 + new_f];
                }
            }
            ++k;
            lds_barrier();
#ifdef DQ_SCAN_STAMPS
            STAMP(t3) prof.t_wait += t1 - t0; prof.t_kloop += t2 - t1; prof.t_tail += t3 - t2; prof.n_acc++;
        } else { STAMP(t1) prof.t_rej += t1 - t0; prof.n_rej++;
#endif
        }
        pos = mypos + 1;
    }
    return true;
}


#define DQ_WALK_REGS __attribute__((amdgpu_num_vgpr(128)))    // allocator confined to v0..v127 and a0..a127: v128..v255 and a128..a223 are the walk's (walk_bodies.inc; checked on the ISA by scripts/check_walk_regs.py)
// ---- the Metropolis walk of one window (n <= 256: one wave per SIMD) ------------------------------------------------------------
// Per accepted flip the walk needs u = G_eff[:, i] and w = G_eff[i, :] - e_i of the effective Green's function
// G_eff = G + sum_m U_m W_m^T (delayed updates): one pass over the k <= 32 pending pairs.  How it is organised, and why (round-3
// stamps and ISA):
//   * decisions for 8 proposals at once (lane q <-> proposal 8 g + q), ballot + first-set jump over the rejections; decisions after
//     the first accepted one are discarded because the diagonal changes, rejected ones before it are exact;
//   * a lane's own pairs {U_m[j], W_m[j]} and the prefetched G column / row elements of three groups of 8 proposals live in
//     registers the allocator never sees (walk_bodies.inc, generated: literal register names, kernels compiled with
//     amdgpu_num_vgpr(128)).  Left to the allocator they were the walk's main cost: as vectors with a dynamic slot index, 24
//     s_set_gpr_idx mode switches + 16 moves + 8 selects per flip; as named scalars behind a dispatch on k, ~70 reconciliation moves
//     at the joins and one prefetch set loaded into temporaries behind s_waitcnt vmcnt(0);
//   * the pass over the pairs is ONE chain of 2 x 32 v_fmac_f64_dpp (pivot entries {U_m[i], W_m[i]}: two compact LDS reads, lane r
//     of every 16-lane row holds pairs r and 16 + r, broadcast by row_newbcast), entered at pair k - 1 by a computed jump: exactly
//     2k FMAs run, nothing is masked, the code exists once;
//   * the pivot's two prefetched G elements are picked lazily, on acceptance only, behind a counted s_waitcnt vmcnt(32) (the two
//     younger groups' 32 loads may still be in flight); the row G[site, :] is read from the transposed copy GT (coalesced);
//   * NO global store inside the walk: panel rows, fields and exp(V) entries are written when the window closes; barriers order
//     LDS only; the diagonal is double-buffered in LDS (one barrier per accepted flip).
#ifndef DQ_W6_KPUB
#define DQ_W6_KPUB 24        // asynchronous windows: pairs per publish ...
#define DQ_W6_XOVER 8        // ... and accepted flips the walk goes on for before it waits for their flush (SliceAsync)
#endif
#define DQ_W6_CAT2(a, b) a##b
#define DQ_W6_CAT(a, b) DQ_W6_CAT2(a, b)
#define DQ_W6_REBASE DQ_W6_CAT(DQ_W6_REBASE_, DQ_W6_KPUB)
#include "walk_bodies.inc"
#pragma clang diagnostic ignored "-Winline-asm"

// issues the loads of the 8 proposals of group g into prefetch set gs = g % 3 (AGPRs; nothing waits here).  Addresses: constant
// scalar bases G / GT + a 32-bit lane offset (j + n * site) * 8, the column part n * site * 8 taken from one lane-distributed LDS
// read of the group's sites by a DPP broadcast inside the add (one instruction per proposal in front of its two loads).
template <bool COH>
__device__ __forceinline__ void walk6_load_group(const double* __restrict__ G, const double* __restrict__ GT, const ScanShared& sh, int g, int gs, int n, int j, bool live) {
    const unsigned jj8 = (unsigned)(live ? j : n - 1) * 8u, n8 = (unsigned)n * 8u;
    const int ig = sh.site[min(g * 8 + (int)(threadIdx.x & 7), n - 1)];       // lane q (mod 8): site of proposal 8 g + q, clamped: loads stay unconditional
    // lane q of every 16-lane row holds the site of proposal q: its column offset reaches all lanes as the DPP operand of the add that
    // forms the address -- one instruction per proposal (v_readlane + scalar multiply + add with their VALU -> SALU -> VALU hops before)
    const unsigned igs = (unsigned)ig * n8;
    unsigned off8[8];
    // (one statement: the s_nop covers the VALU write of igs -> DPP read hazard, which the compiler does not see inside inline asm)
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
        : "=&v"(off8[0]), "=&v"(off8[1]), "=&v"(off8[2]), "=&v"(off8[3]), "=&v"(off8[4]), "=&v"(off8[5]), "=&v"(off8[6]), "=&v"(off8[7])
        : "v"(igs), "v"(jj8));
    if (gs == 0) { if (COH) { DQ_W6_LOADSET_0_COH } else { DQ_W6_LOADSET_0_PLAIN } }
    else if (gs == 1) { if (COH) { DQ_W6_LOADSET_1_COH } else { DQ_W6_LOADSET_1_PLAIN } }
    else { if (COH) { DQ_W6_LOADSET_2_COH } else { DQ_W6_LOADSET_2_PLAIN } }
}

constexpr unsigned SLICE_SPIN_LIMIT = 1u << 22;     // polls (>= 1 us each) a resident partner is given
constexpr unsigned SLICE_CENSUS_SPINS = 256;        // polls the walk grants late flush workgroups before it goes solo (~100-200 us)

__device__ __attribute__((noinline)) void solo_flush(double* G, double* GT, const double2* UW, int n, int k);
// ---- the walk workgroup's side of the persistent kernel's hand-off (protocol: see slice_kernel) ----
// census: has every flush workgroup checked in?  `first` is wave 0's load of the check-in words issued ahead of the caller's drain.
// Returns true when some are still missing after the bounded wait (the walk then goes solo).  Block-wide (two barriers).
__device__ __forceinline__ bool slice_census_missing(SliceSync* sy, unsigned epoch, int F, unsigned first, int* flag) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        unsigned spins = 0; bool all_in = __all(first == slice_tag(epoch, 0));
        while (!all_in) {
            if (++spins > SLICE_CENSUS_SPINS) break;
            __builtin_amdgcn_s_sleep(8);
            const unsigned a = lane < F ? __hip_atomic_load(&sy->arrive[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : slice_tag(epoch, 0);
            all_in = __all(a == slice_tag(epoch, 0));
        }
        if (lane == 0) *flag = all_in ? 0 : 1;
    }
    __syncthreads();
    const bool missing = *flag != 0;
    __syncthreads();
    return missing;
}
// wait until every tile has absorbed window `win` (wave 0 polls, lane f <-> flush workgroup f, F <= 64: one coalesced load per poll)
__device__ __forceinline__ void slice_wait_arrivals(SliceSync* sy, unsigned epoch, unsigned win, int F, int* info) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        unsigned spins = 0; bool broken = false;
        for (;;) {
            const unsigned a = lane < F ? __hip_atomic_load(&sy->arrive[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : slice_tag(epoch, win);
            if (__all(a == slice_tag(epoch, win))) break;
            if (++spins > SLICE_SPIN_LIMIT) { broken = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (broken && lane == 0) { __hip_atomic_store(&sy->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (info) atomicOr(info, 4); }
    }
    __syncthreads();
}

// Asynchronous windows (persistent kernel only).  The walk itself publishes its first DQ_W6_KPUB = 24 pairs and KEEPS WALKING while the
// flush workgroups apply them: its view G_eff = G_old + all pending pairs does not depend on when the 24 land in memory, as long as it
// reads no element of G meanwhile -- it lives on the columns already prefetched from G_old (the rest of this group and the next two)
// and on room for 8 more pairs.  When either runs out, or the slice ends, it waits for the arrivals (by then mostly there), drops the
// 24 applied pairs (registers and LDS rows 24 .. k-1 move down to 0 ..), and restarts the prefetch from the new G.
struct NoAsync { static constexpr bool enabled = false; };
__device__ __forceinline__ bool as_inflight(const NoAsync&) { return false; }
struct SliceAsync {
    static constexpr bool enabled = true;
    SliceSync* sy; unsigned epoch; int F; int* info; int* flag;
    unsigned win = 1;            // number of the next window to be published
    int published = 0;           // pairs handed off inside the walk (the caller's acceptance count)
    bool solo = false;
    bool inflight = false;       // a published window has not been waited for yet
    int g_limit = 0;             // last group whose columns were prefetched before the publish
};

__device__ __forceinline__ bool as_inflight(const SliceAsync& a) { return a.inflight; }

template <bool COH, bool PANELS = true, class AS = NoAsync>
__device__ __forceinline__ int walk_window6(const ScanShared& sh, int& pos, double& dg, int n, int kd, int j, bool live, const double* __restrict__ G,
                                            const double* __restrict__ GT, const UpdateDesc& d, long slice_off, int chain, int8_t* fields_g,
                                            double* __restrict__ Up, double* __restrict__ Wp, AS& as PROF_ARG) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, q8 = lane & 7;
    DQ_W6_RESERVE;
    int par = 0, k = 0;
#ifdef DQ_SCAN_STAMPS
#define DQ_ST(...) __VA_ARGS__
#else
#define DQ_ST(...)
#endif
    // the published window has been applied: forget its pairs.  Block-wide; ends with every wave past a barrier
    auto async_finish = [&]() {
        if constexpr (AS::enabled) {
            DQ_ST(unsigned long long tf0, tf1, tf2; STAMP(tf0))
            slice_wait_arrivals(as.sy, as.epoch, as.win, as.F, as.info);
            DQ_ST(STAMP(tf1))
            DQ_W6_REBASE;
            if (live) for (int t = 0; t < k - DQ_W6_KPUB; ++t) sh.UW[t * n + j] = sh.UW[(DQ_W6_KPUB + t) * n + j];
            if (j < k - DQ_W6_KPUB) sh.acc_site[j] = sh.acc_site[DQ_W6_KPUB + j];
            k -= DQ_W6_KPUB; ++as.win; as.inflight = false;
            lds_barrier();
            DQ_ST(STAMP(tf2) prof.t_finwait += tf1 - tf0; prof.t_fin += tf2 - tf1;)
        }
    };
    int g = pos >> 3;
    bool done = false;
    // no vector-memory operation of this workgroup may be in flight when the counted waits below start counting.  Twice on purpose:
    // the asm is the real wait; the builtin is the one the compiler's waitcnt pass understands -- without it a load of the caller's
    // hand-off loop (a poll whose result it considers pending around the window loop) made it put s_waitcnt vmcnt(0) in front of the
    // pair pass of EVERY accepted flip, i.e. a wait for the 32 prefetch loads just issued (seen in the ISA; +100 ns per flip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);                                        // vmcnt(0), expcnt / lgkmcnt untouched
    walk6_load_group<COH>(G, GT, sh, g, g % 3, n, j, live);
    walk6_load_group<COH>(G, GT, sh, g + 1, (g + 1) % 3, n, j, live);
    while (!done) {
        DQ_ST(unsigned long long tg0, tg1; STAMP(tg0))
        const int gs = g % 3;                                                  // register set of this group
        if constexpr (AS::enabled) {
            if (as.inflight && g > as.g_limit) {                               // out of columns of the old G
                DQ_ST(prof.n_fin_g++;)
                async_finish();
                walk6_load_group<COH>(G, GT, sh, g, gs, n, j, live);
                walk6_load_group<COH>(G, GT, sh, g + 1, (g + 1) % 3, n, j, live);
            }
            if (!as.inflight) walk6_load_group<COH>(G, GT, sh, g + 2, (g + 2) % 3, n, j, live);
        } else
        walk6_load_group<COH>(G, GT, sh, g + 2, (g + 2) % 3, n, j, live);
        DQ_ST(STAMP(tg1) prof.t_first += tg1 - tg0;)
        const int gbase = g * 8;
        // what lane q needs to decide proposal gbase + q is read once per group; only the diagonal changes between passes
        const int p_l = gbase + q8, pc_l = min(p_l, n - 1);
        const int i_l = sh.site[pc_l];
        const double dl_l = sh.dlt[pc_l], rb_l = sh.rbv[pc_l], ur_l = sh.ur[pc_l];
        for (;;) {
            DQ_ST(unsigned long long t0, t1, t2, t3; STAMP(t0))
            // decisions for the (remaining) proposals of the group under the current diagonal: lane q <-> proposal gbase + q
            const double* dcur = par ? sh.diag2 : sh.diag;
            const double r_l = 1.0 + (1.0 - dcur[i_l]) * dl_l;                  // det ratio per flavour (source/model.cpp:95)
            const double R_l = rb_l * (r_l * r_l);                               // :121
            // bernoulli(min(1,|R|)): u < p (source/update.cpp:24, include/utility.h:34-37)
            const bool acc_l = p_l >= pos && p_l < n && ur_l < fmin(1.0, fabs(R_l));
            const unsigned bal = (unsigned)__ballot(acc_l) & 0xffu;
            if (bal == 0u) { pos = min(gbase + 8, n); DQ_ST(STAMP(t1) prof.t_noacc += t1 - t0; prof.n_noacc++;) break; }   // the rest of the group is rejected
            const int first = __ffs((int)bal) - 1;
            const int p = gbase + first;
            pos = p + 1;
            const int i = __builtin_amdgcn_readlane(i_l, first);
            const double dl_p = readlane_f64(dl_l, first), r_p = readlane_f64(r_l, first);
            // pivot entries of the pending pairs, broadcast by DPP in the pass below: lane r16 of every row holds {U_m[i], W_m[i]} for
            // m = r16 (pa) and m = 16 + r16 (pb); entries m >= k are never used.  Read here, in front of the asm statements (their
            // "memory" clobbers pin LDS reads behind them): the LDS latency runs under the wait and the pick
            const int kc = max(k - 1, 0);
            const double2 pa = sh.UW[min(r16, kc) * n + i], pb = sh.UW[min(16 + r16, kc) * n + i];
            const double pref = dl_p / r_p;                                      // source/model.cpp:132 (IEEE division) for the accepted proposal only, issued behind the LDS reads: its dozen dependent operations run under their latency
            // the pivot's G column / row elements from set gs: this group's 16 loads are done once at most the 32 younger ones are pending
            int ulo, uhi, wlo, whi;
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");                   // (while a window is in flight nothing is outstanding at all: its publish drained)
            DQ_W6_PICK(gs * 8 + first);
            double uj = __hiloint2double(uhi, ulo), wj = __hiloint2double(whi, wlo);
            DQ_ST(STAMP(t1))
            {
                // G_eff[:, i], G_eff[i, :]: the pass over the k pending pairs (walk_bodies.inc), pivot entries broadcast by DPP from pa / pb
                double u1 = 0.0, w1 = 0.0;
                DQ_W6_PAIRS;
                uj += u1; wj += w1;
                if (j == i) wj -= 1.0;                                           // V(i) -= 1 (source/model.cpp:135)
                const double pu = pref * uj;
                dg += pu * wj;
                if (live) { sh.UW[k * n + j] = make_double2(pu, wj); (par ? sh.diag : sh.diag2)[j] = dg; }     // slot k / the diagonal buffer no wave is reading
                if (j == 0) sh.acc_site[k] = p;                                  // fields are written at window end
                DQ_W6_SETPAIR(pu, wj);
            }
            DQ_ST(STAMP(t2))
            ++k; par ^= 1;
            lds_barrier();
            DQ_ST(STAMP(t3) prof.t_wait += t1 - t0; prof.t_kloop += t2 - t1; prof.t_tail += t3 - t2; prof.n_acc++;)
            if constexpr (AS::enabled) {
                if (k == DQ_W6_KPUB && kd == UPDATE_KD && !as.inflight && !as.solo && pos < n) {
                    // ---- publish pairs 0 .. 23 and walk on ----
                    DQ_ST(unsigned long long tp0, tp1; STAMP(tp0))
                    if (live) DQ_W6_DUMP_PUB;
                    if (j < DQ_W6_KPUB) {
                        const int pp = sh.acc_site[j];
                        const int ii = sh.site[pp], new_f = sh.newf[pp];
                        fields_g[ii] = (int8_t)new_f;
                        d.expv[(long)chain * d.v_stride + slice_off + ii] = sh.tl[24 + new_f];
                        d.invexpv[(long)chain * d.v_stride + slice_off + ii] = sh.tl[28 + new_f];
                    }
                    unsigned census = slice_tag(as.epoch, 0);
                    if (as.win == 1 && threadIdx.x < (unsigned)as.F) census = __hip_atomic_load(&as.sy->arrive[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // panel stores (and the youngest prefetch loads) of every wave are done
                    __syncthreads();
                    if (as.win == 1 && slice_census_missing(as.sy, as.epoch, as.F, census, as.flag)) {
                        // some flush workgroup is not resident: nobody else will touch G in this launch
                        as.solo = true;
                        if (threadIdx.x == 0) {
                            __hip_atomic_store(&as.sy->seq, ((unsigned long long)slice_tag(as.epoch, as.win) << 32) | SLICE_SOLO_BIT | (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_add(&as.sy->solo_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        solo_flush(const_cast<double*>(G), const_cast<double*>(GT), sh.UW, n, k);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                        as.published += k; k = 0; ++as.win;
                        walk6_load_group<COH>(G, GT, sh, g, gs, n, j, live);
                        walk6_load_group<COH>(G, GT, sh, g + 1, (g + 1) % 3, n, j, live);
                        walk6_load_group<COH>(G, GT, sh, g + 2, (g + 2) % 3, n, j, live);
                    } else {
                        if (threadIdx.x == 0)
                            __hip_atomic_store(&as.sy->seq, ((unsigned long long)slice_tag(as.epoch, as.win) << 32) | (unsigned)DQ_W6_KPUB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        as.published += DQ_W6_KPUB; as.inflight = true; as.g_limit = g + 2;
                    }
                    DQ_ST(STAMP(tp1) prof.t_pub += tp1 - tp0; prof.n_pub++;)
                }
                if (k >= kd || (as.inflight && k >= DQ_W6_KPUB + DQ_W6_XOVER)) {
                    if (!as.inflight) { done = true; break; }
                    DQ_ST(prof.n_fin_k++;)
                    async_finish();                                            // room for pairs again; this group's columns from the new G
                    walk6_load_group<COH>(G, GT, sh, g, gs, n, j, live);
                    walk6_load_group<COH>(G, GT, sh, g + 1, (g + 1) % 3, n, j, live);
                    walk6_load_group<COH>(G, GT, sh, g + 2, (g + 2) % 3, n, j, live);
                }
            } else
            if (k >= kd) { done = true; break; }
        }
        if (pos >= n) done = true;
        ++g;
    }
    if constexpr (AS::enabled) { if (as.inflight) { DQ_ST(prof.n_fin_e++;) async_finish(); } }
#undef DQ_ST
#ifdef DQ_SCAN_STAMPS
    unsigned long long td0; STAMP(td0)
#endif
    // (prefetch loads may still be in flight into their AGPR sets: the caller drains vmcnt before anything else can use them)
    // ---- window end: the flush's operands and the accepted field changes leave the workgroup (coalesced, from registers) ----
    if (PANELS && live) DQ_W6_DUMP;
    if (j < k) {
        const int p = sh.acc_site[j];
        const int i = sh.site[p], new_f = sh.newf[p];
        fields_g[i] = (int8_t)new_f;                                        // field.set_single_field (source/update.cpp:28)
        d.expv[(long)chain * d.v_stride + slice_off + i] = sh.tl[24 + new_f];
        d.invexpv[(long)chain * d.v_stride + slice_off + i] = sh.tl[28 + new_f];
    }
#ifdef DQ_SCAN_STAMPS
    { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); unsigned long long td1; STAMP(td1) prof.t_dump += td1 - td0; }
#endif
    return k;
}

__device__ __forceinline__ void scan_shared_init(ScanShared& sh, unsigned char* smem, int n, int kd, bool regs) {
    sh.UW = reinterpret_cast<double2*>(smem);
    sh.diag = reinterpret_cast<double*>(sh.UW + (size_t)kd * n);
    sh.dlt = sh.diag + n; sh.rbv = sh.dlt + n; sh.ur = sh.rbv + n; sh.tl = sh.ur + n;
    sh.site = reinterpret_cast<int*>(sh.tl + 32);
    sh.newf = reinterpret_cast<signed char*>(sh.site + n);
    if (regs) { sh.diag2 = reinterpret_cast<double*>(smem + scan_lds_bytes(n, kd, false)); sh.acc_site = reinterpret_cast<int*>(sh.diag2 + n); }
}

// MAXT: the launch bound.  For n <= 256 the kernel runs one wave per SIMD and may use the whole 512-entry register
// file (walk v3); under a 1024-thread bound (128 VGPRs) only the LDS variant fits.
template <int MAXT>
__global__ __launch_bounds__(MAXT) DQ_WALK_REGS void scan_kernel(UpdateDesc d, int l, int acc_slot, int first, int kd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int chain = blockIdx.y;
    const int n = d.n;
    const int j = threadIdx.x;
    const bool live = j < n;
    constexpr bool REGS = MAXT <= 256;

    ScanShared sh;
    scan_shared_init(sh, smem, n, kd, REGS);

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
    if constexpr (REGS) {
        const double* __restrict__ GT = d.GT.at(chain);
        double dg0 = live ? sh.diag[j] : 0.0;
        { NoAsync na; k = walk_window6<false>(sh, pos, dg0, n, kd, j, live, G, GT, d, slice_off, chain, fields_g, Up, Wp, na PROF_PASS); }
    } else {
        double pcA[SCAN_PF], prA[SCAN_PF], pcB[SCAN_PF], prB[SCAN_PF];
        int g = pos / SCAN_PF;
        scan_prefetch<false>(pcA, prA, G, sh, g, n, j, live);
        scan_prefetch<false>(pcB, prB, G, sh, g + 1, n, j, live);
        for (;;) {
            if (!scan_group<false>(pcA, prA, sh, g, n, kd, j, live, pos, k, d, slice_off, chain, fields_g, Up, Wp PROF_PASS)) break;
            scan_prefetch<false>(pcA, prA, G, sh, g + 2, n, j, live);
            if (!scan_group<false>(pcB, prB, sh, g + 1, n, kd, j, live, pos, k, d, slice_off, chain, fields_g, Up, Wp PROF_PASS)) break;
            scan_prefetch<false>(pcB, prB, G, sh, g + 3, n, j, live);
            g += 2;
        }
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

// One wave's share of a flush: G[a,b] += sum_{m<k} U_m[a] W_m[b] on its 16x16 sub-tile, and (WITH_GT) the mirrored
// sub-tile of the transposed copy GT[b,a].  Latency matters more than flops here (N x k x N with k <= 32): the G tile
// and the panel rows are all loaded in ONE round trip (panel rows unconditionally for all KD slots -- stale slots are
// masked after the fact).  The second MFMA chain (operands swapped) produces the transposed tile in the layout whose
// stores are coalesced, instead of scattering the first chain's accumulators.
// PAIRED: the panels are stored as P[m >> 1][a][m & 1] (what the register walk's window-end dump writes with 16-byte stores);
// otherwise as P[m][a] (the LDS walk of n > 256, which stores rows as it goes).
template <bool COH, bool WITH_GT, bool PAIRED>
__device__ __forceinline__ void flush_tile(double* __restrict__ G, double* __restrict__ GT, const double* __restrict__ Up, const double* __restrict__ Wp,
                                           int a0, int b0, int n, int k, int kd, int lane) {
    const int r = lane & 15, kk = lane >> 4;
    const int a = min(a0 + r, n - 1), b = min(b0 + r, n - 1);          // clamped: loads stay unconditional
    const bool a_ok = a0 + r < n, b_ok = b0 + r < n;
    double uv[UPDATE_KD / 4], wv[UPDATE_KD / 4], gv[4], gt[4];
#pragma unroll
    for (int s = 0; s < UPDATE_KD / 4; ++s) {
        const int m = min(4 * s + kk, kd - 1);
        const int ib = PAIRED ? (m >> 1) * 2 * n + 2 * b + (m & 1) : m * n + b, ia = PAIRED ? (m >> 1) * 2 * n + 2 * a + (m & 1) : m * n + a;
        if (COH) { wv[s] = ld_coh(Wp + ib); uv[s] = ld_coh(Up + ia); }
        else { wv[s] = Wp[ib]; uv[s] = Up[ia]; }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int bb = min(b0 + kk + 4 * reg, n - 1), aa = min(a0 + kk + 4 * reg, n - 1);
        if (COH) { gv[reg] = ld_coh(G + a + (long)n * bb); if (WITH_GT) gt[reg] = ld_coh(GT + b + (long)n * aa); }
        else { gv[reg] = G[a + (long)n * bb]; if (WITH_GT) gt[reg] = GT[b + (long)n * aa]; }
    }
    if (k == 0) return;
    d4 acc = {0.0, 0.0, 0.0, 0.0}, acc_t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < UPDATE_KD / 4; ++s) {
        const bool ok = 4 * s + kk < k;
        const double wm = (ok && b_ok) ? wv[s] : 0.0, um = (ok && a_ok) ? uv[s] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wm, um, acc, 0, 0, 0);                     // A: rows <-> b, B: cols <-> a
        if (WITH_GT) acc_t = __builtin_amdgcn_mfma_f64_16x16x4f64(um, wm, acc_t, 0, 0, 0);    // A: rows <-> a, B: cols <-> b
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int bb = b0 + kk + 4 * reg, aa = a0 + kk + 4 * reg;
        if (a_ok && bb < n) { if (COH) st_coh(G + a + (long)n * bb, gv[reg] + acc[reg]); else G[a + (long)n * bb] = gv[reg] + acc[reg]; }
        if (WITH_GT && b_ok && aa < n) { if (COH) st_coh(GT + b + (long)n * aa, gt[reg] + acc_t[reg]); else GT[b + (long)n * aa] = gt[reg] + acc_t[reg]; }
    }
}

// one block per 32x32 tile, 2x2 waves
template <bool WITH_GT>
__global__ __launch_bounds__(256) void flush_kernel(UpdateDesc d, int tiles_per_dim, int kd) {
    const int chain = blockIdx.y;
    const int n = d.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a0 = (blockIdx.x % tiles_per_dim) * 32 + (wave & 1) * 16;
    const int b0 = (blockIdx.x / tiles_per_dim) * 32 + (wave >> 1) * 16;
    if (a0 >= n || b0 >= n) return;
    // the walk of a slice usually ends after 4 of the 8 launched windows: an empty window must not touch G at all
    // (batched engines are bandwidth-bound here: 40 % of the flush traffic was for windows with nothing to apply)
    const int k = __builtin_amdgcn_readfirstlane(d.state[(long)chain * d.state_stride + 1]);
    if (k == 0) return;
    flush_tile<false, WITH_GT, WITH_GT>(d.G.at(chain), WITH_GT ? d.GT.at(chain) : nullptr, d.Upanel + (long)chain * d.panel_stride, d.Wpanel + (long)chain * d.panel_stride,
                               a0, b0, n, k, kd, lane);
}

// ---- solo slice kernel: one workgroup per chain does the walk AND its own flushes --------------------------------------
// For engines with many chains the single-launch kernel above cannot be used (its 1 + (n/32)^2 workgroups per chain must all
// be resident), and the scan / flush kernel pairs cost 2 x ceil(N/KD) launches per slice with every chain waiting for the
// slowest one twice per window.  Here a chain needs nobody else: after each window its workgroup applies G += U^T W and
// GT += W^T U itself, the MFMA operands taken straight from the LDS pair store (no panel round trip through memory), 64
// 16x16 sub-tiles per wave.  A flush costs ~27 us on one CU (fp64 MFMA bound) instead of a ~5 us hand-off, which is the wrong
// trade for ONE chain and the right one once there are as many chains as CUs: every CU works on its own chain and the slice is
// one launch (measured break-even between 128 and 256 chains, see launch_update_slice).
// NB sub-tiles at a time: all their G / GT loads are issued before the first MFMA (a sub-tile alone is one L2 round trip per 1024 MFMA
// clocks).  Plain loads and stores: only this workgroup touches the chain during the launch, and workgroup-scope synchronisation
// (barrier) is all its waves need to see each other's stores through the CU's own L1.
template <int NB>
__device__ __forceinline__ void flush_tiles_lds(double* __restrict__ G, double* __restrict__ GT, const double2* __restrict__ UW, int st0, int st_step,
                                                int n_st, int tiles16, int n, int k, int lane) {
    const int r = lane & 15, kk = lane >> 4;
    double gv[NB][4], gt[NB][4];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int st = min(st0 + q * st_step, n_st - 1);
        const int a0 = (st % tiles16) * 16, b0 = (st / tiles16) * 16;
        const int a = min(a0 + r, n - 1), b = min(b0 + r, n - 1);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int bb = min(b0 + kk + 4 * reg, n - 1), aa = min(a0 + kk + 4 * reg, n - 1);
            gv[q][reg] = G[a + (long)n * bb]; gt[q][reg] = GT[b + (long)n * aa];
        }
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int st = st0 + q * st_step;
        if (st >= n_st) break;                                        // wave-uniform
        const int a0 = (st % tiles16) * 16, b0 = (st / tiles16) * 16;
        const int a = min(a0 + r, n - 1), b = min(b0 + r, n - 1);
        const bool a_ok = a0 + r < n, b_ok = b0 + r < n;
        d4 acc = {0.0, 0.0, 0.0, 0.0}, acc_t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < UPDATE_KD / 4; ++s) {
            if (4 * s >= k) break;                                    // wave-uniform
            const int m = 4 * s + kk;
            const bool ok = m < k;
            const int mc = min(m, k - 1);
            const double um = (ok && a_ok) ? UW[mc * n + a].x : 0.0, wm = (ok && b_ok) ? UW[mc * n + b].y : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wm, um, acc, 0, 0, 0);                 // rows <-> b, cols <-> a
            acc_t = __builtin_amdgcn_mfma_f64_16x16x4f64(um, wm, acc_t, 0, 0, 0);             // rows <-> a, cols <-> b
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int bb = b0 + kk + 4 * reg, aa = a0 + kk + 4 * reg;
            if (a_ok && bb < n) G[a + (long)n * bb] = gv[q][reg] + acc[reg];
            if (b_ok && aa < n) GT[b + (long)n * aa] = gt[q][reg] + acc_t[reg];
        }
    }
}

// ---- persistent slice kernel: the whole local update of one time slice in ONE launch --------------------------------
// grid = (1 + F, chains): workgroup 0 of a chain runs the serial Metropolis walk (window after window); the other
// F = (n/32)^2 workgroups each own one 32x32 tile of G (and its mirror in GT) and apply every window's low-rank
// correction to it.  Replaces ceil(n/KD) x (scan launch + flush launch) per slice: per window the cost of two kernel
// boundaries, a cold prologue and an empty-launch tail becomes one publish / arrive hand-off.
//
// Hand-off (SliceSync in common.h; cdna_hip_programming.md Guideline 16, form R1).  Every shared byte -- panel rows, G / GT tiles,
// the words -- is written and read with agent-scope relaxed atomics (global_store / global_load ... sc1: write-through, served
// past the CU's L1), ONE wave per workgroup polls, arrivals are per-workgroup words read by the walk with one coalesced load:
//   walk  -> flush : when a window closes the panel rows are stored, every storing wave drains (s_waitcnt vmcnt(0): an agent-scope
//                    store that has retired is performed at agent scope -- the same property the architected store-release
//                    sequence "buffer_wbl2 sc1; s_waitcnt vmcnt(0); store sc1" relies on for the stores in front of it), the
//                    workgroup barriers, one lane stores seq = {tag, k | flags}.  Flush workgroups poll that ONE word; their
//                    payload loads are issued after the poll has matched (branch on the loaded value) and a workgroup barrier.
//   flush -> walk  : tiles are read and written the same way; each flush wave drains, barrier, one lane stores the tag into the
//                    workgroup's arrival word; the walk polls the F words (lane f <-> word f) until all carry the tag, barriers,
//                    then restarts its prefetch (the diagonal continues from the lane's running value).
//   No plain load ever touches a byte another workgroup writes during the launch, so no acquire fence (buffer_inv sc1: drops the
//   L1 copies a plain load could hit) is needed; no plain store carries payload, so no release fence (buffer_wbl2 sc1: writes back
//   dirty L2 lines) is needed.  scripts/handoff_litmus.hip checks exactly these instruction forms word by word under load.
// Tags carry the launch number (slice_tag), so no word is re-armed and no word of an earlier launch can match.
// Residency is CHECKED, not assumed: every flush workgroup checks in (arrive[f] = tag of window 0) before it polls; before the
// walk publishes its FIRST window -- before it has modified anything but its own registers and LDS -- it looks for all F
// check-ins (they arrived long ago: the first window takes ~30 us) and, when some are still missing after a bounded wait,
// publishes the SOLO flag instead: flush workgroups that are or become resident leave without touching G, and the walk
// workgroup applies every window itself from its LDS pair store (flush_tiles_lds, the solo kernel's flush).  Once all have
// checked in, every later hand-off completes in bounded time (resident workgroups always make progress), so the spin bounds
// below are a net for device faults only.
// the solo fall-back's flush as a real function call: inlined, its registers join the walk's allocation problem and the hot loop of
// the walk ends up with tuple copies and scratch spills (seen in the ISA); a call happens between windows, where almost nothing is live.
// A called function is NOT bound by the kernel's register budget (its ISA uses v128 and up freely) and the compiler saves nothing of the
// hand-managed file around the call: every call site sits where that file is dead -- all pending pairs have just been applied (k = 0
// afterwards) and the prefetch sets are reloaded from the new G right after it.
__device__ __attribute__((noinline)) void solo_flush(double* G, double* GT, const double2* UW, int n, int k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles16 = (n + 15) / 16, n_st = tiles16 * tiles16;
    for (int st = wave; st < n_st; st += 4 * 4) flush_tiles_lds<4>(G, GT, UW, st, 4, n_st, tiles16, n, k, lane);
}

// NFIX / KFIX: matrix size and window depth as compile-time constants (0 = taken from the arguments).  The headline size N = 256, kd = 32
// gets its own instance: the LDS layout, the strides of the pair store and the tile grid become literals, which frees the scalar
// registers the generic instance spills to VGPR lanes in the walk's loops (53 spilled SGPRs, ~1.8 us per slice)
template <int NFIX, int KFIX>
__global__ __launch_bounds__(256) DQ_WALK_REGS void slice_kernel(UpdateDesc d, SliceSync* sync_p, int l, int acc_slot, int kd_arg, int tiles_arg, int* info) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int chain = blockIdx.y;
    const int n = NFIX ? NFIX : d.n;
    const int kd = KFIX ? KFIX : kd_arg;
    const int tiles_per_dim = NFIX ? (NFIX + 31) / 32 : tiles_arg;
    const int F = tiles_per_dim * tiles_per_dim;
    SliceSync* sy = sync_p + chain;
    double* __restrict__ G = d.G.at(chain);
    double* __restrict__ GT = d.GT.at(chain);
    double* __restrict__ Up = d.Upanel + (long)chain * d.panel_stride;
    double* __restrict__ Wp = d.Wpanel + (long)chain * d.panel_stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned epoch = d.slice_epoch;

    if (blockIdx.x > 0) {
        // ================= flush role: one 32x32 tile of G, one 16x16 sub-tile per wave =================
        const int tile = blockIdx.x - 1;
        if (tile == d.slice_absent_tile && (d.slice_absent_l < 0 || d.slice_absent_l == l)) return;                      // debug: a workgroup that never becomes resident
        if (tile == d.slice_late_tile) {                              // debug: a workgroup that becomes resident LATE (after the walk's census has given up)
            const unsigned long long t0 = wall_clock64();             // 100 MHz
            while (wall_clock64() - t0 < 100ull * (unsigned long long)d.slice_late_us) __builtin_amdgcn_s_sleep(64);
        }
        const int a0 = (tile % tiles_per_dim) * 32 + (wave & 1) * 16;
        const int b0 = (tile / tiles_per_dim) * 32 + (wave >> 1) * 16;
        if (t == 0) __hip_atomic_store(&sy->arrive[tile], slice_tag(epoch, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // census: resident
        // one wave polls the window word (64 workgroups x 4 waves polling one L2 line slow the walk's own loads: guide, Pitfall 9),
        // the others wait at the barrier and take the word from LDS
        unsigned long long* bcast = reinterpret_cast<unsigned long long*>(smem);
        for (unsigned win = 1;; ++win) {
            if (wave == 0) {
                unsigned long long word = 0; unsigned spins = 0; bool give_up = false;
                for (;;) {
                    word = __hip_atomic_load(&sy->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned tag = (unsigned)(word >> 32);
                    if (tag == slice_tag(epoch, win)) break;
                    if ((tag >> 8) == (slice_tag(epoch, 0) >> 8)) {
                        // a word of THIS launch for another window.  Solo: the walk gave up on the census before this workgroup checked in
                        // and has been applying its windows itself ever since (every word it publishes carries the bit) -- a workgroup that
                        // becomes resident late sees window 2, 3, ... and never window 1: it leaves, G is not its to touch.  Anything else
                        // (a later window without the solo bit) cannot happen while this workgroup's arrival is outstanding: fail at once.
                        if ((unsigned)word & SLICE_SOLO_BIT) break;
                        if ((tag & 0xffu) > win) { give_up = true; break; }
                    }
                    if (++spins > SLICE_SPIN_LIMIT) { give_up = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (give_up && lane == 0) { __hip_atomic_store(&sy->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (info) atomicOr(info, 4); }
                if (lane == 0) bcast[win & 1] = give_up ? ~0ULL : word;
            }
            __syncthreads();
            const unsigned long long word = bcast[win & 1];
            if (word == ~0ULL) break;
            const unsigned lo = (unsigned)word;
            if (lo & SLICE_SOLO_BIT) break;                           // the walk applies the windows itself: G is not ours to touch
            const int k = (int)(lo & 0x3fffffffu);
            const bool final = (lo & SLICE_FINAL_BIT) != 0;
            if (k > 0 && a0 < n && b0 < n) flush_tile<true, true, true>(G, GT, Up, Wp, a0, b0, n, k, kd, lane);
            if (final) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains before the arrival is signalled
            __syncthreads();
            // arrival: this workgroup's own word (no read-modify-write on a shared counter: 64 atomics on one word serialise in L2)
            if (t == 0) __hip_atomic_store(&sy->arrive[tile], slice_tag(epoch, win), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // ================= walk role =================
        const int j = t;
        const bool live = j < n;
        ScanShared sh;
        scan_shared_init(sh, smem, n, kd, true);
        const double* tab_g = reinterpret_cast<const double*>(d.tabs + chain);
        const long slice_off = (long)l * n;
        int8_t* fields_g = d.fields + (long)chain * d.f_stride + slice_off;
        double dg = 0.0;
        if (live) {
            const long off = (long)chain * d.rs_stride + slice_off + j;
            const int i = d.perm[off];
            const int kp = d.kprop[off];
            const int old_f = fields_g[i];
            sh.site[j] = i; sh.newf[j] = (signed char)c_proposal[old_f][kp];
            sh.rbv[j] = tab_g[old_f * 3 + kp]; sh.dlt[j] = tab_g[12 + old_f * 3 + kp]; sh.ur[j] = d.u[off];
            dg = G[j + (long)n * j];                                  // first window: G was written by the previous kernel
            sh.diag[j] = dg;
        }
        if (j < 32) sh.tl[j] = tab_g[j];
        __syncthreads();
        int pos = 0, total_acc = 0;
        int* flag = sh.acc_site + UPDATE_KD;                          // spare LDS word behind acc_site: wave 0's verdict on the census
#ifdef DQ_SCAN_STAMPS
        ScanProf prof; unsigned long long tk0, t_hand = 0; STAMP(tk0)
#endif
        SliceAsync as{sy, epoch, F, info, flag};
        for (;; ++as.win) {
            const int k = walk_window6<true, true, SliceAsync>(sh, pos, dg, n, kd, j, live, G, GT, d, slice_off, chain, fields_g, Up, Wp, as PROF_PASS);
            total_acc += k + as.published; as.published = 0;
            const unsigned win = as.win;
            bool solo = as.solo;
            const bool final = pos >= n;
#ifdef DQ_SCAN_STAMPS
            unsigned long long th0; STAMP(th0)
#endif
            // census (first window): is every flush workgroup resident?  One coalesced load, issued in front of the drain of the panel
            // stores so that its round trip hides behind theirs (the workgroups checked in while the first window was walked)
            unsigned census = slice_tag(epoch, 0);
            if (win == 1 && wave == 0 && lane < F) census = __hip_atomic_load(&sy->arrive[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // panel stores of every wave have left the CU
            __syncthreads();
            if (win == 1) { solo = slice_census_missing(sy, epoch, F, census, flag); as.solo = solo; }
            if (t == 0) {
                __hip_atomic_store(&sy->seq, ((unsigned long long)slice_tag(epoch, win) << 32) | (final ? SLICE_FINAL_BIT : 0u) | (solo ? SLICE_SOLO_BIT : 0u) | (unsigned)k,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (solo && win == 1) __hip_atomic_fetch_add(&sy->solo_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (solo) {
                // nobody else touches G during this launch: apply the window from the LDS pair store (plain accesses, this CU's L1 is
                // write-through; the walk's own sc1 loads are served by the L2 the stores went to)
                if (k > 0) solo_flush(G, GT, sh.UW, n, k);
                if (final) break;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (live) sh.diag[j] = dg;
                __syncthreads();
                continue;
            }
            if (final) break;
            slice_wait_arrivals(sy, epoch, win, F, info);                // until every tile has absorbed this window
            // G changed: the prefetch restarts from memory.  The diagonal does not: the lane's running value dg (advanced by pu * wj at
            // every accepted flip) IS G_jj after the flush up to the summation order, so the next window starts from it instead of
            // paying an L2 round trip in front of the prefetch (it is re-read from G at the start of every slice)
            if (live) sh.diag[j] = dg;
            __syncthreads();
#ifdef DQ_SCAN_STAMPS
            { unsigned long long th1; STAMP(th1) t_hand += th1 - th0; }
#endif
        }
        if (j == 0) d.acc_out[(long)chain * d.acc_stride + acc_slot] = total_acc;
#ifdef DQ_SCAN_STAMPS
        if (j == 0) { unsigned long long tk1; STAMP(tk1)
#if 1
            printf("slice l=%d: total %llu cyc | hand-offs %llu | %d acc: decide+fetch %llu pairs %llu tail %llu | window start->first flip %llu | %d empty passes %llu | dump+drain %llu | %d async publishes %llu, finishes (full %d / columns %d / end %d): wait %llu rebase %llu\n",
                   l, tk1 - tk0, t_hand, prof.n_acc, prof.t_wait, prof.t_kloop, prof.t_tail, prof.t_first, prof.n_noacc, prof.t_noacc, prof.t_dump,
                   prof.n_pub, prof.t_pub, prof.n_fin_k, prof.n_fin_g, prof.n_fin_e, prof.t_finwait, prof.t_fin);
#endif
        }
#endif
    }
}

// resume = 1: finish a slice that scan / flush kernel pairs have walked up to state[0] (exits at once when nothing is left)
__global__ __launch_bounds__(256) DQ_WALK_REGS void slice_solo_kernel(UpdateDesc d, int l, int acc_slot, int kd, int resume) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int chain = blockIdx.y;
    const int n = d.n;
    double* __restrict__ G = d.G.at(chain);
    double* __restrict__ GT = d.GT.at(chain);
    const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
    const bool live = j < n;
    int pos = resume ? d.state[(long)chain * d.state_stride] : 0;
    if (pos >= n) return;                                             // wave-uniform: the pairs already finished this chain's slice
    ScanShared sh;
    scan_shared_init(sh, smem, n, kd, true);
    const double* tab_g = reinterpret_cast<const double*>(d.tabs + chain);
    const long slice_off = (long)l * n;
    int8_t* fields_g = d.fields + (long)chain * d.f_stride + slice_off;
    double dg = 0.0;
    if (live) {
        // (resume: sites are visited once per slice, so the fields of the proposals still ahead are the ones the slice started with)
        const long off = (long)chain * d.rs_stride + slice_off + j;
        const int i = d.perm[off];
        const int kp = d.kprop[off];
        const int old_f = fields_g[i];
        sh.site[j] = i; sh.newf[j] = (signed char)c_proposal[old_f][kp];
        sh.rbv[j] = tab_g[old_f * 3 + kp]; sh.dlt[j] = tab_g[12 + old_f * 3 + kp]; sh.ur[j] = d.u[off];
        dg = G[j + (long)n * j];
        sh.diag[j] = dg;
    }
    if (j < 32) sh.tl[j] = tab_g[j];
    __syncthreads();
    int total_acc = 0;
    const int tiles16 = (n + 15) / 16;
#ifdef DQ_SCAN_STAMPS
    ScanProf prof;
#endif
    for (;;) {
        NoAsync na; const int k = walk_window6<false, false>(sh, pos, dg, n, kd, j, live, G, GT, d, slice_off, chain, fields_g, nullptr, nullptr, na PROF_PASS);
        total_acc += k;
        __syncthreads();                                              // every pair of the window is in LDS
        if (k > 0) {
            const int n_st = tiles16 * tiles16;                       // 16x16 sub-tiles of G (and their mirrors in GT): wave w takes w, w + 4, ...
            for (int st = wave; st < n_st; st += 4 * 8) flush_tiles_lds<8>(G, GT, sh.UW, st, 4, n_st, tiles16, n, k, lane);
        }
        if (pos >= n) break;
        __syncthreads();                                              // fence + barrier: the flushed tiles are visible to every wave of this workgroup
        if (live) { dg = G[j + (long)n * j]; sh.diag[j] = dg; }
        __syncthreads();
    }
    if (j == 0) { int* acc = d.acc_out + (long)chain * d.acc_stride + acc_slot; *acc = (resume ? *acc : 0) + total_acc; }
}

static int pick_kd(int n) {
    // LDS budget: 16*kd*n + n*(5*8+4+1) + tables <= ~150 KiB
    const long budget = 150 * 1024 - (long)n * 48 - 1024;
    long kd = budget / (16L * n);
    if (kd > UPDATE_KD) kd = UPDATE_KD;
    if (kd < 1) kd = 1;
    return (int)kd;
}

// ---- co-residency of the persistent slice kernel ------------------------------------------------------------------------
// slice_kernel's 1 + (n/32)^2 workgroups per chain wait for each other, so all of them must be resident at once, and each
// occupies a whole CU (512 VGPRs per lane: one wave per SIMD).  Engines therefore RESERVE their workgroups against the
// device's CU count when they are created (hipDeviceProp_t::multiProcessorCount minus a margin for whatever else runs);
// an engine that does not get its reservation -- many chains per engine, many engines per device -- takes the
// scan / flush kernel pairs, which need nobody to be co-resident.  Kernels that do not spin (GEMMs, factorisations) finish
// on their own and only delay a hand-off.  The reservation is per process: several PROCESSES sharing one GPU with
// persistent engines should set DQMC_SLICE_MULTIKERNEL=1 (a hand-off that times out is reported, never a hang).
static std::mutex g_resv_mu;
static int g_resv_used[64] = {};
static int g_resv_cap[64] = {};
int slice_workgroups(int n, int n_chains) { return (1 + slice_flush_workgroups(n)) * n_chains; }
bool slice_reserve(int device, int n, int n_chains) {
    if (device < 0 || device >= 64 || n > 1024) return false;
    std::lock_guard<std::mutex> lk(g_resv_mu);
    if (g_resv_cap[device] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
        const char* m = getenv("DQMC_SLICE_CU_MARGIN");
        g_resv_cap[device] = prop.multiProcessorCount - (m ? atoi(m) : 32);
    }
    const int want = slice_workgroups(n, n_chains);
    if (g_resv_used[device] + want > g_resv_cap[device]) return false;
    g_resv_used[device] += want;
    return true;
}
void slice_release(int device, int n, int n_chains) {
    if (device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lk(g_resv_mu);
    g_resv_used[device] -= slice_workgroups(n, n_chains);
}

// kernels with more than 64 KiB of dynamic LDS need the attribute on every device they run on (init_device_kernels, engine.hip)
int update_init_device() {
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(scan_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(scan_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(slice_kernel<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(slice_kernel<256, UPDATE_KD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(slice_solo_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

int launch_update_slice(const UpdateDesc& d, int l, int acc_slot, int n_chains, hipStream_t s, int* gt_kept) {
    const int n = d.n;
    if (gt_kept) *gt_kept = 1;
    if (n > 1024) { set_error("local update kernel supports n_sites <= 1024"); return -1; }
    const int kd = pick_kd(n);
    const int threads = ((n + 63) / 64) * 64;
    const bool regs = threads <= 256;                               // walk v3 (register-resident pending pairs, needs GT)
    const size_t lds = scan_lds_bytes(n, kd, regs);
    const int tiles = (n + 31) / 32;
    if (regs) {
        if (!d.GT.p) { set_error("local update: transposed workspace missing"); return -1; }
        if (!d.gt_valid) { if (int rc = launch_transpose_scale(CMat(d.G.p, d.G.stride), d.GT, CVec(), n, n_chains, s)) return rc; }      // GT = G^T, kept in step by the flushes
    }
    static const bool multi_kernel = getenv("DQMC_SLICE_MULTIKERNEL") != nullptr;      // A/B switch
    // persistent single-launch path: needs every workgroup of a chain resident at once, one per CU (the walk's LDS)
    // sub-matrix updates (update_sm.hip): the same Markov chain from k x k algebra per proposal instead of 2 n k flops per accepted flip.
    // Opt-in: measured 211 us per cfg-3 slice against 138 us for the delayed-update walk below -- the k x k algebra is a dependent chain
    // on one wave per SIMD (860 clk per two-proposal pass, 1780 clk per accepted flip), see DESIGN.md
    static const bool submatrix_walk = getenv("DQMC_WALK_SUBMATRIX") != nullptr;
    if (!multi_kernel && d.slice_sync && d.Cpanel && d.GT.p && (submatrix_walk || !regs)) {       // n > 256: the default (see update_sm.hip)
        if (!regs && !d.gt_valid) { if (int rc = launch_transpose_scale(CMat(d.G.p, d.G.stride), d.GT, CVec(), n, n_chains, s)) return rc; }
        return launch_update_slice_sm(d, l, acc_slot, n_chains, s);
    }
    if (!multi_kernel && regs && d.slice_sync) {                    // slice_sync is only handed out with a CU reservation (slice_reserve)
        if (n == 256 && kd == UPDATE_KD)
            hipLaunchKernelGGL((slice_kernel<256, UPDATE_KD>), dim3(1 + tiles * tiles, n_chains), dim3(256), lds, s, d, reinterpret_cast<SliceSync*>(d.slice_sync), l, acc_slot, kd,
                               tiles, d.info);
        else
            hipLaunchKernelGGL((slice_kernel<0, 0>), dim3(1 + tiles * tiles, n_chains), dim3(256), lds, s, d, reinterpret_cast<SliceSync*>(d.slice_sync), l, acc_slot, kd,
                               tiles, d.info);
        DQ_HIP(hipGetLastError());
        return 0;
    }
    // measured (cfg 3, sweeps/s, solo vs pairs): 64 chains 166 / 214, 128 chains 251 / 275, 256 chains 324 / 307 -- a chain's own CU
    // flushes slower than the whole chip does, so the solo kernel pays once there are enough chains to occupy every CU
    if (regs && !multi_kernel && n_chains >= 224) {
        hipLaunchKernelGGL(slice_solo_kernel, dim3(1, n_chains), dim3(256), lds, s, d, l, acc_slot, kd, 0);
        DQ_HIP(hipGetLastError());
        return 0;
    }
    int windows = (n + kd - 1) / kd;
    // a thermalised slice ends after ~4 windows; the scan / flush pairs past that point are launches that find nothing to do (8 of 16
    // at cfg 3).  Four pairs, then ONE solo launch that finishes whatever a chain has left (usually nothing: it exits at once).
    const bool tail_solo = regs && !multi_kernel && windows > 4;
    if (tail_solo) windows = 4;
    for (int w = 0; w < windows; ++w) {
        if (regs) {
            hipLaunchKernelGGL(scan_kernel<256>, dim3(1, n_chains), dim3(threads), lds, s, d, l, acc_slot, w == 0 ? 1 : 0, kd);
            hipLaunchKernelGGL(flush_kernel<true>, dim3(tiles * tiles, n_chains), dim3(256), 0, s, d, tiles, kd);
        } else {
            if (gt_kept) *gt_kept = 0;                                                 // flush_kernel<false> updates G only
            hipLaunchKernelGGL(scan_kernel<1024>, dim3(1, n_chains), dim3(threads), lds, s, d, l, acc_slot, w == 0 ? 1 : 0, kd);
            hipLaunchKernelGGL(flush_kernel<false>, dim3(tiles * tiles, n_chains), dim3(256), 0, s, d, tiles, kd);
        }
    }
    if (tail_solo) {
        hipLaunchKernelGGL(slice_solo_kernel, dim3(1, n_chains), dim3(256), lds, s, d, l, acc_slot, kd, 1);
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
