// update_sm.hip -- update::local_update (source/update.cpp:5-32) for one time slice with SUB-MATRIX updates: the
// second of the two speed-ups the reference's README lists as future work ("delayed / sub-matrix updates",
// README.md:41).  Persistent single-launch kernel for n <= 1024 and few chains.  It is the default
// for 256 < n <= 1024 (the per-proposal cost of the k x k algebra does not grow with n, where the delayed update's
// 2 n k flops per accepted flip and its scan / flush launch pairs do: 0.44 ms against 1.65 ms per cfg-5 slice) and opt-in
// (DQMC_WALK_SUBMATRIX=1) at n <= 256, where the register-resident delayed-update walk of update.hip is faster; batched
// engines keep the delayed-update kernels.
//
// Within a window of up to KD = 32 accepted flips at sites S = (s_1 .. s_k) the Green's function is never touched.
// Every accepted flip is a rank-1 change (AttractiveHubbard::update_greens_local, source/model.cpp:124-138) whose
// column factor lies in span G0[:, S] and whose row factor lies in span (G0[S, :] - E_S^T), so
//
//     G_k = G0 + A C_k B^T,      A = G0[:, S]  (n x k),   B^T = G0[S, :] - E_S^T  (k x n),   C_k  k x k.
//
// The acceptance ratio of a proposal at site j (each site is visited once per slice, so j is not in S) needs
//     G_k[j, j] = G0[j, j] + alpha^T C_k beta,      alpha = A[j, :] = G0[j, S],   beta = B[j, :] = G0[S, j],
// i.e. k^2 flops on the k x k matrix instead of the 2 n k flops per accepted flip of the delayed update, and an
// accepted flip extends C by one row and column:
//     u~ = [C beta; 1],  w~ = [C^T alpha; 1],  C <- [[C, 0], [0, 0]] + pref u~ w~^T,   pref = delta / (1 + (1 - G_jj) delta).
// When the window closes,  G += (A C) B^T  is applied by the flush workgroups on the matrix cores.  The Markov chain
// is the reference's; only the rounding order differs (1e-14 against the sequential rank-1 form, tests/test_oracle.py).
//
// Walk workgroup (256 threads, one wave per SIMD, thread t <-> sites t, t + 256, ... : NS = ceil(n / 256) site slots):
//   * LDS holds A^T and B^T of the window, restricted to the sites STILL TO BE VISITED and indexed by visiting position:
//     cols[m][q] = G0[site(pos0 + q), s_m], rows[m][q] = G0[s_m, site(pos0 + q)] - [..], written by all threads when flip m
//     is accepted from the column / row elements every thread prefetches for every proposal (as in the delayed-update
//     walk; G0 does not change inside a window, so the prefetch is exact).  A window that starts later in the slice
//     needs fewer columns and holds more flips (struct SmShared below);
//   * every wave keeps its own copy of C and C^T in registers (lane m of each 32-lane half holds row m) and evaluates
//     the SAME two proposals per pass (half 0: proposal pos, half 1: proposal pos + 1): y = C beta and z = C^T alpha with
//     the operands broadcast by the DPP network (v_fmac_f64_dpp row_newbcast), alpha^T y by a DPP reduction over the half.  All
//     waves take identical decisions, so the only synchronisation is one LDS barrier per accepted flip (the cols / rows
//     slot of the flip must be complete before the next pass reads it);
//   * the full column / row of an accepted flip (the flush's operands A^T, B^T) go to the panel buffers when the flip is
//     accepted (st_coh_opaque), C at the window's end; nothing else is stored to memory inside a window.
// Flush workgroups (one or two 32 x 32 tiles of G and of its transposed copy GT each, one 16 x 16 sub-tile per wave):
//   D1 = C^T A^T (matrix cores), whose accumulator registers are directly the operands of the two final chains
//   G_tile += D1^T B^T and GT_tile += B D1 (an MFMA D tile is the next product's A / B operand without a shuffle).
// Hand-off protocol: the one of slice_kernel in update.hip (tagged window word, arrival counter, exit ticket).
#include "common.h"
#include <cstdlib>
#include "wave.h"

namespace dq {

namespace {

using d4 = __attribute__((ext_vector_type(4))) double;

__constant__ int c_proposal_sm[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};   // include/field.h:45-48

constexpr int SM_KD = UPDATE_KD;               // largest window: 2 x 16 broadcast slots per 16-lane row
constexpr unsigned SM_SPIN_LIMIT = 1u << 22;      // polls (>= 1 us each) a resident partner is given
constexpr unsigned SM_CENSUS_SPINS = 1024;        // polls the walk grants late flush workgroups at kernel start (~0.5 ms) before it leaves the slice untouched
// LDS row stride of cols / rows (doubles), = 1 mod 32: rows m = 0..15 of one column land in distinct banks
__host__ __device__ inline int sm_ls(int n) { return ((n + 31) / 32) * 32 + 1; }

__device__ __forceinline__ double ld_coh(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_coh(double* p, double x) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same store as st_coh, written as inline asm: the panel stores of an accepted flip sit inside the walk's pass loop, and a loop with
// VMEM stores and no VMEM loads makes the compiler's wait-count pass flush vmcnt in the loop preheader (SIInsertWaitcnts, "flush in
// preheader") -- right after the next group's prefetch has been issued: a full memory round trip per group of 8 proposals (~150 us of a
// 665 us cfg-5 slice).  Stores the pass does not see only make its later vmcnt(N) waits stricter than needed (the counter retires in
// order), never unsafe; the window end drains them with an explicit s_waitcnt vmcnt(0) before the hand-off word is published.
__device__ __forceinline__ void st_coh_opaque(double* p, double x) {
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void lds_barrier() {              // orders LDS traffic only (no vmcnt(0): the prefetch stays in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// rows 2 hx and 2 hx + 1 (16 lanes each) of x replicated into all four rows of the wave: ra <- row 2 hx, rb <- row 2 hx + 1 (hx wave-uniform).
// gfx950 v_permlane32_swap / v_permlane16_swap, two VALU instructions per dword, instead of ds_bpermute through the LDS crossbar.
__device__ __forceinline__ void sm_bcast_rows(double x, int hx, double& ra, double& rb) {
    const unsigned lo = (unsigned)__double_as_longlong(x), hi = (unsigned)(__double_as_longlong(x) >> 32);
    const auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);       // [0]: lower half in both halves, [1]: upper half
    const auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const unsigned vlo = hx ? l32[1] : l32[0], vhi = hx ? h32[1] : h32[0];
    const auto l16 = __builtin_amdgcn_permlane16_swap(vlo, vlo, false, false);     // [0]: even row of each pair in both, [1]: odd row
    const auto h16 = __builtin_amdgcn_permlane16_swap(vhi, vhi, false, false);
    ra = __hiloint2double((int)h16[0], (int)l16[0]); rb = __hiloint2double((int)h16[1], (int)l16[1]);
}

struct SmShared {
    double* panels;                             // cols | rows of the current window, [kd_w][ls_w] each, re-laid-out per window (below)
    long room;                                  // doubles available for the two panels
    double* diag0; double* dlt; double* rbv; double* ur; double* tl;
    int* site; signed char* newf; int* acc_site; int* posof;
};
// LDS: the fixed part first, then the panels.  Within a window only the sites still to be visited are ever looked up, so the panels are
// indexed by POSITION in the slice's visiting order relative to the window's first proposal (column q <-> position pos0 + q): a window that
// starts at position pos0 needs n - pos0 columns, and the window size grows as the slice proceeds,
//     kd_w = min(32, room / (2 ls_w)),   ls_w = sm_ls(n - pos0)
// (n = 576: 15, 16, 17, 19, 21, 24, 28, 32, ... instead of 15 throughout: 10-11 windows per slice instead of 16, each ~27 us of hand-off).
__host__ __device__ inline size_t sm_fixed_bytes(int n) { return (size_t)n * 32 + 256 + (size_t)n * 8 + (((size_t)n + 7) & ~(size_t)7) + SM_KD * 4 + 64; }
__host__ __device__ inline long sm_room_doubles(int n) { return (160L * 1024 - 256 - (long)sm_fixed_bytes(n)) / 8; }
__host__ __device__ inline size_t sm_lds_bytes(int n) { return sm_fixed_bytes(n) + (size_t)sm_room_doubles(n) * 8; }
__host__ __device__ inline int sm_window_kd(long room, int ls) {
    const long kd = room / (2L * ls);
    return (int)(kd > SM_KD ? SM_KD : kd < 1 ? 1 : kd);
}
__device__ __forceinline__ void sm_shared_init(SmShared& sh, unsigned char* smem, int n) {
    sh.diag0 = reinterpret_cast<double*>(smem);
    sh.dlt = sh.diag0 + n; sh.rbv = sh.dlt + n; sh.ur = sh.rbv + n; sh.tl = sh.ur + n;
    sh.site = reinterpret_cast<int*>(sh.tl + 32);
    sh.posof = sh.site + n;
    sh.newf = reinterpret_cast<signed char*>(sh.posof + n);
    sh.acc_site = reinterpret_cast<int*>(sh.newf + ((n + 7) & ~7));
    sh.panels = reinterpret_cast<double*>(smem + sm_fixed_bytes(n));
    sh.room = sm_room_doubles(n);
}

// G[tt, site] (PC##q##_sl) and G[site, tt] (PR##q##_sl, from the transposed copy) of the 8 proposals of group gg for the thread's site
// slots sl (tt = t + 256 sl).  The values of a group are SCALARS named by token pasting, not arrays: the accepted proposal's pair is selected by a switch, and with arrays
// the optimiser folds that switch into one dynamically indexed load, which pins all three register sets in scratch.
#define SM_DECL1(P, q) double P##q##_0 = 0.0, P##q##_1 = 0.0, P##q##_2 = 0.0, P##q##_3 = 0.0
#define SM_DECL8(P) SM_DECL1(P, 0); SM_DECL1(P, 1); SM_DECL1(P, 2); SM_DECL1(P, 3); SM_DECL1(P, 4); SM_DECL1(P, 5); SM_DECL1(P, 6); SM_DECL1(P, 7)
#define SM_LOADS(PC, PR, q, sl)                                                                                     \
    {                                                                                                               \
        const unsigned off8_ = (unsigned)(tt##sl + n * i_) * 8u;                                                    \
        PC##q##_##sl = ld_coh(reinterpret_cast<const double*>(reinterpret_cast<const char*>(G) + off8_));           \
        PR##q##_##sl = ld_coh(reinterpret_cast<const double*>(reinterpret_cast<const char*>(GT) + off8_));          \
    }
#define SM_LOAD1(PC, PR, gg, q)                                                                                     \
    {                                                                                                               \
        const int i_ = sh.site[min((gg) * 8 + q, n - 1)];             /* clamped: loads stay unconditional */       \
        SM_LOADS(PC, PR, q, 0)                                                                                      \
        if constexpr (NS > 1) SM_LOADS(PC, PR, q, 1)                                                                \
        if constexpr (NS > 2) SM_LOADS(PC, PR, q, 2)                                                                \
        if constexpr (NS > 3) SM_LOADS(PC, PR, q, 3)                                                                \
    }
#define SM_LOAD8(PC, PR, gg)                                                                                        \
    SM_LOAD1(PC, PR, gg, 0) SM_LOAD1(PC, PR, gg, 1) SM_LOAD1(PC, PR, gg, 2) SM_LOAD1(PC, PR, gg, 3)                 \
    SM_LOAD1(PC, PR, gg, 4) SM_LOAD1(PC, PR, gg, 5) SM_LOAD1(PC, PR, gg, 6) SM_LOAD1(PC, PR, gg, 7)
#define SM_PICK1(PC, PR, q) gc0 = PC##q##_0; gr0 = PR##q##_0; gc1 = PC##q##_1; gr1 = PR##q##_1; gc2 = PC##q##_2; gr2 = PR##q##_2; gc3 = PC##q##_3; gr3 = PR##q##_3;
#define SM_PICK8(PC, PR)                                                                                            \
    switch (first) {                                                                                                \
        case 0: SM_PICK1(PC, PR, 0) break; case 1: SM_PICK1(PC, PR, 1) break;                                       \
        case 2: SM_PICK1(PC, PR, 2) break; case 3: SM_PICK1(PC, PR, 3) break;                                       \
        case 4: SM_PICK1(PC, PR, 4) break; case 5: SM_PICK1(PC, PR, 5) break;                                       \
        case 6: SM_PICK1(PC, PR, 6) break; default: SM_PICK1(PC, PR, 7) break;                                      \
    }

// acc_a += bcast(pv, lane R0) * M[MB] + bcast(pv, R2) * M[MB + 2];  acc_b += bcast(pv, R1) * M[MB + 1] + bcast(pv, R3) * M[MB + 3]
// (pv: lane r of every 16-lane row holds entry r of the broadcast operand).  The hazard recogniser does not look inside
// inline asm: the block opens with the wait states a DPP read of a just-written VGPR needs.
#define SM_DOT4(R0, R1, R2, R3, acc_a, acc_b, pv, M, MB)                                                            \
    asm("s_nop 1\n\t"                                                                                               \
        "v_fmac_f64_dpp %0, %2, %3 row_newbcast:" #R0 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %1, %2, %4 row_newbcast:" #R1 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %0, %2, %5 row_newbcast:" #R2 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %1, %2, %6 row_newbcast:" #R3 " row_mask:0xf bank_mask:0xf"                                 \
        : "+v"(acc_a), "+v"(acc_b)                                                                                  \
        : "v"(pv), "v"(M[MB]), "v"(M[MB + 1]), "v"(M[MB + 2]), "v"(M[MB + 3]))
// M[MB + e] += bcast(pv, lane R_e) * x  for e = 0..3
#define SM_AXPY4(R0, R1, R2, R3, M, MB, pv, x)                                                                      \
    asm("s_nop 1\n\t"                                                                                               \
        "v_fmac_f64_dpp %0, %4, %5 row_newbcast:" #R0 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %1, %4, %5 row_newbcast:" #R1 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %2, %4, %5 row_newbcast:" #R2 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %3, %4, %5 row_newbcast:" #R3 " row_mask:0xf bank_mask:0xf"                                 \
        : "+v"(M[MB]), "+v"(M[MB + 1]), "+v"(M[MB + 2]), "+v"(M[MB + 3])                                            \
        : "v"(pv), "v"(x))
// the two products of a pass, y = C beta and z = C^T alpha, interleaved: four independent accumulator chains, a dependent fp64 FMA only every
// fourth instruction (the z product used to follow the decision, on the accepted path: ~300 clk of dependent chain per accepted flip)
#define SM_DOT4X2(R0, R1, R2, R3, ya, yb, za, zb, pvy, pvz, M, N, MB)                                               \
    asm("s_nop 1\n\t"                                                                                               \
        "v_fmac_f64_dpp %0, %4, %6 row_newbcast:" #R0 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %1, %4, %7 row_newbcast:" #R1 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %2, %5, %10 row_newbcast:" #R0 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %3, %5, %11 row_newbcast:" #R1 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %0, %4, %8 row_newbcast:" #R2 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %1, %4, %9 row_newbcast:" #R3 " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %2, %5, %12 row_newbcast:" #R2 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %3, %5, %13 row_newbcast:" #R3 " row_mask:0xf bank_mask:0xf"                                  \
        : "+v"(ya), "+v"(yb), "+v"(za), "+v"(zb)                                                                    \
        : "v"(pvy), "v"(pvz), "v"(M[MB]), "v"(M[MB + 1]), "v"(M[MB + 2]), "v"(M[MB + 3]),                           \
          "v"(N[MB]), "v"(N[MB + 1]), "v"(N[MB + 2]), "v"(N[MB + 3]))
#define SM_DOT8X2LO(ya, yb, za, zb, pvy, pvz, M, N) SM_DOT4X2(0, 1, 2, 3, ya, yb, za, zb, pvy, pvz, M, N, 0); SM_DOT4X2(4, 5, 6, 7, ya, yb, za, zb, pvy, pvz, M, N, 4)
#define SM_DOT8X2HI(ya, yb, za, zb, pvy, pvz, M, N) SM_DOT4X2(8, 9, 10, 11, ya, yb, za, zb, pvy, pvz, M, N, 8); SM_DOT4X2(12, 13, 14, 15, ya, yb, za, zb, pvy, pvz, M, N, 12)
// y = C v, z = C^T w over the first k columns: rows in (c0 | c1) and (t0 | t1), v broadcast from (va, vb), w from (wa_, wb_)
#define SM_MATVEC2(outy, outz, va, vb, wa_, wb_)                                                                    \
    {                                                                                                               \
        double y0_ = 0.0, y1_ = 0.0, z0_ = 0.0, z1_ = 0.0;                                                          \
        if (k > 0) { SM_DOT8X2LO(y0_, y1_, z0_, z1_, va, wa_, c0, t0); }                                            \
        if (k > 8) { SM_DOT8X2HI(y0_, y1_, z0_, z1_, va, wa_, c0, t0); }                                            \
        if (k > 16) { SM_DOT8X2LO(y0_, y1_, z0_, z1_, vb, wb_, c1, t1); }                                           \
        if (k > 24) { SM_DOT8X2HI(y0_, y1_, z0_, z1_, vb, wb_, c1, t1); }                                           \
        outy = y0_ + y1_; outz = z0_ + z1_;                                                                         \
    }
#define SM_DOT16(acc_a, acc_b, pv, M)                                                                               \
    SM_DOT4(0, 1, 2, 3, acc_a, acc_b, pv, M, 0); SM_DOT4(4, 5, 6, 7, acc_a, acc_b, pv, M, 4);                       \
    SM_DOT4(8, 9, 10, 11, acc_a, acc_b, pv, M, 8); SM_DOT4(12, 13, 14, 15, acc_a, acc_b, pv, M, 12)
#define SM_DOT8LO(acc_a, acc_b, pv, M) SM_DOT4(0, 1, 2, 3, acc_a, acc_b, pv, M, 0); SM_DOT4(4, 5, 6, 7, acc_a, acc_b, pv, M, 4)
#define SM_DOT8HI(acc_a, acc_b, pv, M) SM_DOT4(8, 9, 10, 11, acc_a, acc_b, pv, M, 8); SM_DOT4(12, 13, 14, 15, acc_a, acc_b, pv, M, 12)
#define SM_AXPY8LO(M, pv, x) SM_AXPY4(0, 1, 2, 3, M, 0, pv, x); SM_AXPY4(4, 5, 6, 7, M, 4, pv, x)
#define SM_AXPY8HI(M, pv, x) SM_AXPY4(8, 9, 10, 11, M, 8, pv, x); SM_AXPY4(12, 13, 14, 15, M, 12, pv, x)

// y = M v over the first k columns (8 at a time): M row in (m0: columns 0..15, m1: 16..31), v broadcast from (va, vb)
#define SM_MATVEC(out, m0, m1, va, vb)                                                                              \
    {                                                                                                               \
        double s0_ = 0.0, s1_ = 0.0;                                                                                \
        if (k > 0) { SM_DOT8LO(s0_, s1_, va, m0); }                                                                 \
        if (k > 8) { SM_DOT8HI(s0_, s1_, va, m0); }                                                                 \
        if (k > 16) { SM_DOT8LO(s0_, s1_, vb, m1); }                                                                \
        if (k > 24) { SM_DOT8HI(s0_, s1_, vb, m1); }                                                                \
        out = s0_ + s1_;                                                                                            \
    }

#ifdef DQ_SM_STAMPS
#define SMT(v) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); v = _t; }
struct SmProf { unsigned long long t_pass = 0, t_acc = 0, t_dump = 0, t_hand = 0, t_first = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0; int n_pass = 0, n_acc = 0; };
#define SM_PROF_ARG , SmProf& prof
#define SM_PROF_PASS , prof
#define SMX(...) __VA_ARGS__
#else
#define SM_PROF_ARG
#define SM_PROF_PASS
#define SMX(...)
#endif

// Walks one window: from proposal `pos` until KD flips are pending or the slice is over; then writes A^T, B^T, C (the
// flush's operands) and the accepted field changes to memory.  Returns the number of flips.
template <int NS>
__device__ __forceinline__ int sm_walk_window(const SmShared& sh, int& pos, int n, int t, const double* __restrict__ G,
                                              const double* __restrict__ GT, const UpdateDesc& d, long slice_off, int chain, int8_t* fields_g,
                                              double* __restrict__ Ap, double* __restrict__ Bp, double* __restrict__ Cp, bool reload_diag SM_PROF_ARG) {
    const int lane = t & 63, wave = t >> 6, h = lane >> 5, m = lane & 31, r16 = lane & 15;
    // after a flush the diagonal restarts from memory: its loads are issued first and reach LDS only after the first prefetch sets
    // have been requested as well, so that the two round trips overlap (they used to follow each other: 1.2 us per window)
    double dgl[NS];
    if (reload_diag) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) { const int tt = min(t + 256 * s_, n - 1); dgl[s_] = ld_coh(G + tt + (long)n * tt); }
    }
    SMX(unsigned long long tw0; SMT(tw0))
    double c0[16], c1[16], t0[16], t1[16];                        // C[m][0..15], C[m][16..31], C^T[m][0..15], C^T[m][16..31] (static indices only: registers)
    SM_DECL8(pcA); SM_DECL8(prA); SM_DECL8(pcB); SM_DECL8(prB); SM_DECL8(pcC); SM_DECL8(prC);   // prefetched G column / row elements of group (g % 3)
    const int tt0 = min(t, n - 1), tt1 = min(t + 256, n - 1), tt2 = min(t + 512, n - 1), tt3 = min(t + 768, n - 1);   // clamped site slots
    (void)tt1; (void)tt2; (void)tt3;
    // this window's panels: columns <-> positions pos0, pos0 + 1, ... (the sites still to be visited)
    const int pos0 = pos;
    const int LS = sm_ls(n - pos0);
    const int kd = sm_window_kd(sh.room, LS);
    double* const cols = sh.panels;
    double* const rows = sh.panels + (long)kd * LS;
    // panel column of the thread's own sites (negative: visited before this window, never looked up again)
    const int q0 = sh.posof[tt0] - pos0, q1 = sh.posof[tt1] - pos0, q2 = sh.posof[tt2] - pos0, q3 = sh.posof[tt3] - pos0;
    (void)q1; (void)q2; (void)q3;
#pragma unroll
    for (int e = 0; e < 16; ++e) { c0[e] = 0.0; c1[e] = 0.0; t0[e] = 0.0; t1[e] = 0.0; }
    int k = 0;
    int g = pos >> 3;
    bool done = false;
    // NS <= 2: three register sets, the loads of groups g + 1 and g + 2 in flight while group g is walked; NS >= 3 (n > 512): two
    // sets (a third would not fit the register file), the loads of group g + 1 in flight -- a group takes >= 4 passes = 1.4 us
    constexpr int LEAD = NS <= 2 ? 2 : 1;
    if constexpr (NS <= 2) {
        switch (g % 3) {
            case 0: SM_LOAD8(pcA, prA, g) SM_LOAD8(pcB, prB, g + 1) break;      /* the sets of groups g and g + 1; the group macro requests g + 2 when it starts on g */
            case 1: SM_LOAD8(pcB, prB, g) SM_LOAD8(pcC, prC, g + 1) break;
            default: SM_LOAD8(pcC, prC, g) SM_LOAD8(pcA, prA, g + 1) break;
        }
    } else {
        if (g % 2 == 0) { SM_LOAD8(pcA, prA, g) } else { SM_LOAD8(pcB, prB, g) }
    }
    if (reload_diag) {                                            // workgroup-uniform
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) { const int tt = t + 256 * s_; if (tt < n) sh.diag0[tt] = dgl[s_]; }
        lds_barrier();
    }
#define SM_SLOT(sl)                                                                                                                      \
    if (t + 256 * sl < n) {                                                                                                              \
        const double cv_ = gc##sl, rv_ = (t + 256 * sl == i) ? gr##sl - 1.0 : gr##sl;                                                    \
        if (q##sl >= 0) { cols[k * LS + q##sl] = cv_; rows[k * LS + q##sl] = rv_; }                                                      \
        st_coh_opaque(Ap + k * n + t + 256 * sl, cv_); st_coh_opaque(Bp + k * n + t + 256 * sl, rv_);                                                  \
    }
#define SM_GROUP(PCS, PRS, PCT, PRT)                                                                                                     \
    {                                                                                                                                    \
        SM_LOAD8(PCT, PRT, g + LEAD)                                                                                                     \
        const int gend = min(g * 8 + 8, n);                                                                                              \
        while (pos < gend) {                                                                                                             \
            /* two proposals per pass: half 0 <-> pos, half 1 <-> pos + 1 (inside the group, so the prefetch set is the same) */         \
            SMX(unsigned long long t0_, t1_, t2_; SMT(t0_))                                                                              \
            const bool two = pos + 1 < gend;                                                                                             \
            const int pme = (h && two) ? pos + 1 : pos;                                                                                  \
            const int jme = sh.site[pme];                                                                                                \
            const int kc = max(k - 1, 0);                                                                                                \
            /* beta = rows[:, j] and alpha = cols[:, j] in broadcast layout (lane r of every 16-lane row: entries r and 16 + r), alpha_m own */ \
            const int o_a = min(r16, kc) * LS + (pme - pos0), o_b = min(16 + r16, kc) * LS + (pme - pos0);                            \
            double be_a = rows[o_a], be_b = rows[o_b], al_a = cols[o_a], al_b = cols[o_b];                                               \
            const double d0 = sh.diag0[jme], dl = sh.dlt[pme], rb = sh.rbv[pme], uu = sh.ur[pme];                                        \
            if (r16 >= k) { be_a = 0.0; al_a = 0.0; }                                                                                    \
            if (16 + r16 >= k) { be_b = 0.0; al_b = 0.0; }                                                                               \
            const double al_m = (lane & 16) ? al_b : al_a;                   /* alpha_m, m = lane & 31 */                                 \
            double y, z;                                                     /* y_m = sum_n C[m][n] beta_n, z_m = sum_n C[n][m] alpha_n */ \
            SM_MATVEC2(y, z, be_a, be_b, al_a, al_b)                                                                                     \
            double s = al_m * y;                                                                                                         \
            s = row16_sum(s);                                                                                                            \
            s += dpp_mov_f64<0x142, 0xa>(s);                                 /* row_bcast:15: rows 1 and 3 hold the sums of their halves */ \
            const double gjj = d0 + s;                                       /* G_k[j, j] */                                              \
            const double r = 1.0 + (1.0 - gjj) * dl;                         /* det ratio per flavour (source/model.cpp:95) */            \
            /* delta / r (source/model.cpp:132), needed only if the proposal is accepted, formed here next to the decision: v_rcp_f64 + two Newton */ \
            /* steps and a residual correction (an ulp or two from the IEEE quotient) -- eight operations that issue under the latency of the */ \
            /* decision's own chain instead of ~350 clk of dependent chain on the accepted path */                                        \
            double ir_ = __builtin_amdgcn_rcp(r);                                                                                        \
            ir_ = fma(fma(-r, ir_, 1.0), ir_, ir_);                                                                                      \
            ir_ = fma(fma(-r, ir_, 1.0), ir_, ir_);                                                                                      \
            double prf_ = dl * ir_;                                                                                                      \
            prf_ = fma(fma(-r, prf_, dl), ir_, prf_);                                                                                    \
            const double R = rb * (r * r);                                   /* :121 */                                                  \
            const bool acc_l = uu < fmin(1.0, fabs(R));                      /* bernoulli(min(1,|R|)): u < p (source/update.cpp:24) */    \
            const unsigned long long bal = __ballot(acc_l);                                                                              \
            const bool accA = (bal >> 31) & 1ULL, accB = two && ((bal >> 63) & 1ULL);                                                    \
            SMX(SMT(t1_) prof.t_pass += t1_ - t0_; prof.n_pass++;)                                                                       \
            if (!accA && !accB) { pos += two ? 2 : 1; continue; }                                                                        \
            const int hx = accA ? 0 : 1;                                     /* the earlier proposal wins; the later one is re-evaluated */ \
            const int p = pos + hx;                                                                                                      \
            const int first = p - g * 8;                                                                                                 \
            pos = p + 1;                                                                                                                 \
            const int i = __builtin_amdgcn_readlane(jme, hx * 32 + 31);                                                                  \
            const double pref = readlane_f64(prf_, hx * 32 + 31);                                                                        \
            SMX(unsigned long long ta1_, ta2_, ta3_; SMT(ta1_) prof.a1 += ta1_ - t1_;)                                                   \
            /* u~ = [y; 1], w~ = [z; 1] of the accepted half, into the broadcast layout of every row */                                  \
            double ua, ub, wa, wb;                                                                                                       \
            sm_bcast_rows(y, hx, ua, ub); sm_bcast_rows(z, hx, wa, wb);                                                                  \
            if (r16 == (k & 15)) { if (k < 16) { ua = 1.0; wa = 1.0; } else { ub = 1.0; wb = 1.0; } }                                    \
            const double pu = pref * ((lane & 16) ? ub : ua), pw = pref * ((lane & 16) ? wb : wa);                                       \
            /* C[m][n] += pref u~_m w~_n,  C^T[m][n] += pref w~_m u~_n  for n <= k */                                                    \
            SM_AXPY8LO(c0, wa, pu); SM_AXPY8LO(t0, ua, pw);                                                                              \
            if (k >= 8) { SM_AXPY8HI(c0, wa, pu); SM_AXPY8HI(t0, ua, pw); }                                                              \
            if (k >= 16) { SM_AXPY8LO(c1, wb, pu); SM_AXPY8LO(t1, ub, pw); }                                                             \
            if (k >= 24) { SM_AXPY8HI(c1, wb, pu); SM_AXPY8HI(t1, ub, pw); }                                                             \
            SMX(asm volatile("" :: "v"(c0[0]), "v"(t0[0])); SMT(ta2_) prof.a2 += ta2_ - ta1_;)                                           \
            /* slot k of A^T / B^T: the column and the row of G0 at the flipped site, every thread its own element */                   \
            double gc0, gr0, gc1, gr1, gc2, gr2, gc3, gr3;                                                                               \
            SM_PICK8(PCS, PRS)                                                                                                           \
            /* V(i) -= 1 (source/model.cpp:135).  LDS: the sites still to come, by position; memory: the whole column / row, the flush's operands */ \
            SM_SLOT(0) if constexpr (NS > 1) SM_SLOT(1) if constexpr (NS > 2) SM_SLOT(2) if constexpr (NS > 3) SM_SLOT(3)                \
            SMX(SMT(ta3_) prof.a3 += ta3_ - ta2_;)                                                                                       \
            if (t == 0) sh.acc_site[k] = p;                                                                                              \
            ++k;                                                                                                                         \
            lds_barrier();                                                                                                               \
            SMX(SMT(t2_) prof.t_acc += t2_ - t1_; prof.a4 += t2_ - ta3_; prof.n_acc++;)                                                  \
            if (k >= kd) { done = true; break; }                                                                                      \
        }                                                                                                                                \
        if (pos >= n) done = true;                                                                                                       \
        ++g;                                                                                                                             \
    }
    while (!done) {
        if constexpr (NS <= 2) {
            switch (g % 3) {
                case 0: SM_GROUP(pcA, prA, pcC, prC) if (done) break; [[fallthrough]];
                case 1: SM_GROUP(pcB, prB, pcA, prA) if (done) break; [[fallthrough]];
                default: SM_GROUP(pcC, prC, pcB, prB) break;
            }
        } else {
            if (g % 2 == 0) { SM_GROUP(pcA, prA, pcB, prB) if (done) break; }
            SM_GROUP(pcB, prB, pcA, prA)
        }
    }
#undef SM_GROUP
#undef SM_SLOT
    SMX(unsigned long long td0; SMT(td0))
    // ---- window end: C (from the registers of wave 0) leaves the workgroup; A^T and B^T went to memory flip by flip ----
    if (wave == 0 && lane < 32) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { st_coh(Cp + m * SM_KD + e, c0[e]); st_coh(Cp + m * SM_KD + 16 + e, c1[e]); }
    }
    if (t < k) {
        const int p = sh.acc_site[t];
        const int i = sh.site[p], new_f = sh.newf[p];
        fields_g[i] = (int8_t)new_f;                                        // field.set_single_field (source/update.cpp:28)
        d.expv[(long)chain * d.v_stride + slice_off + i] = sh.tl[24 + new_f];
        d.invexpv[(long)chain * d.v_stride + slice_off + i] = sh.tl[28 + new_f];
    }
    SMX({ asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); unsigned long long td1; SMT(td1) prof.t_dump += td1 - td0; prof.t_first += td0 - tw0; })
    return k;
}
static_assert(SM_KD == 32, "the sub-matrix walk keeps C as 2 x 16 register columns per lane");
#undef SM_DECL1
#undef SM_DECL8

// One wave's share of a flush: its 16 x 16 sub-tile of G (rows a0.., columns b0..) and the mirrored sub-tile of GT.
//   D1[nn][a] = sum_m C[m][nn] A^T[m][a]      two 16 x 16 accumulators (nn < 16, nn >= 16), 8 k-steps each
//   G[a][b]  += sum_nn D1[nn][a] B^T[nn][b]    accumulator rows <-> b: D1's registers are the B operands
//   GT[b][a] += the same                      accumulator rows <-> a: D1's registers are the A operands
// Loads and arithmetic are separate calls: a workgroup that owns two tiles (n = 576) requests the operands of both before it computes
// on the first -- one memory round trip per window instead of two in a row.
struct SmFlushOps {
    double av[8], bv[8], gv[4], gt[4];
    int a, b, a0, b0; bool a_ok, b_ok;
};
__device__ __forceinline__ void sm_flush_load_c(const double* __restrict__ Cp, int lane, double (&cv0)[8], double (&cv1)[8]) {
    const int r = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) { const int mm = 4 * s + kk; cv0[s] = ld_coh(Cp + mm * SM_KD + r); cv1[s] = ld_coh(Cp + mm * SM_KD + 16 + r); }
}
__device__ __forceinline__ void sm_flush_load(SmFlushOps& o, const double* __restrict__ G, const double* __restrict__ GT, const double* __restrict__ Ap,
                                              const double* __restrict__ Bp, int a0, int b0, int n, int lane) {
    const int r = lane & 15, kk = lane >> 4;
    o.a0 = a0; o.b0 = b0;
    o.a = min(a0 + r, n - 1); o.b = min(b0 + r, n - 1);                // clamped: loads stay unconditional
    o.a_ok = a0 + r < n; o.b_ok = b0 + r < n;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int mm = 4 * s + kk;                                     // window slot of this lane at k-step s
        o.av[s] = ld_coh(Ap + mm * n + o.a); o.bv[s] = ld_coh(Bp + mm * n + o.b);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int bb = min(b0 + kk + 4 * reg, n - 1), aa = min(a0 + kk + 4 * reg, n - 1);
        o.gv[reg] = ld_coh(G + o.a + (long)n * bb); o.gt[reg] = ld_coh(GT + o.b + (long)n * aa);
    }
}
__device__ __forceinline__ void sm_flush_apply(const SmFlushOps& o, const double (&cv0)[8], const double (&cv1)[8], double* __restrict__ G, double* __restrict__ GT,
                                               int n, int k, int lane) {
    const int kk = lane >> 4;
    d4 d1lo = {0.0, 0.0, 0.0, 0.0}, d1hi = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const bool ok = 4 * s + kk < k;                                // panel rows and C rows beyond k hold stale data
        const double am = (ok && o.a_ok) ? o.av[s] : 0.0, c_lo = ok ? cv0[s] : 0.0, c_hi = ok ? cv1[s] : 0.0;
        d1lo = __builtin_amdgcn_mfma_f64_16x16x4f64(c_lo, am, d1lo, 0, 0, 0);       // rows <-> nn (0..15), columns <-> a
        d1hi = __builtin_amdgcn_mfma_f64_16x16x4f64(c_hi, am, d1hi, 0, 0, 0);       // rows <-> nn (16..31)
    }
    d4 acc = {0.0, 0.0, 0.0, 0.0}, acc_t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const bool ok = 4 * s + kk < k;
        const double bm = (ok && o.b_ok) ? o.bv[s] : 0.0;
        const double dm = s < 4 ? d1lo[s & 3] : d1hi[s & 3];           // D1[nn = 4 s + kk][a = r]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bm, dm, acc, 0, 0, 0);           // rows <-> b, columns <-> a
        acc_t = __builtin_amdgcn_mfma_f64_16x16x4f64(dm, bm, acc_t, 0, 0, 0);       // rows <-> a, columns <-> b
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int bb = o.b0 + kk + 4 * reg, aa = o.a0 + kk + 4 * reg;
        if (o.a_ok && bb < n) st_coh(G + o.a + (long)n * bb, o.gv[reg] + acc[reg]);
        if (o.b_ok && aa < n) st_coh(GT + o.b + (long)n * aa, o.gt[reg] + acc_t[reg]);
    }
}

}  // namespace

// grid = (1 + Fwg, chains): workgroup 0 walks, workgroups 1 .. Fwg flush; flush workgroup f owns the 32 x 32 tiles f, f + Fwg, ...
// of G and GT (one tile each while tiles^2 <= 192; two each at n = 576)
template <int NS>
__global__ __launch_bounds__(256) void slice_sm_kernel(UpdateDesc d, SliceSync* sync_p, int l, int acc_slot, int tiles_per_dim, int* info) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int chain = blockIdx.y;
    const int n = d.n;
    const int F = (int)gridDim.x - 1;                                 // flush workgroups
    const int n_tiles = tiles_per_dim * tiles_per_dim;
    SliceSync* sy = sync_p + chain;
    double* __restrict__ G = d.G.at(chain);
    double* __restrict__ GT = d.GT.at(chain);
    double* __restrict__ Ap = d.Upanel + (long)chain * d.panel_stride;
    double* __restrict__ Bp = d.Wpanel + (long)chain * d.panel_stride;
    double* __restrict__ Cp = d.Cpanel + (long)chain * (SM_KD * SM_KD);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned epoch = d.slice_epoch;                             // hand-off words carry the launch number: SliceSync, common.h

    if (blockIdx.x > 0) {
        // ================= flush role: 32x32 tiles of G, one 16x16 sub-tile per wave =================
        if ((int)blockIdx.x - 1 == d.slice_absent_tile && (d.slice_absent_l < 0 || d.slice_absent_l == l)) return;       // debug: a workgroup that never becomes resident
        if (t == 0) __hip_atomic_store(&sy->arrive[blockIdx.x - 1], slice_tag(epoch, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // census: resident
        // one wave polls the window word, the others take it from LDS (polling cost: guide, Guideline 16 Pitfall 9)
        unsigned long long* bcast = reinterpret_cast<unsigned long long*>(smem);
        for (unsigned win = 1;; ++win) {
            if (wave == 0) {
                unsigned long long w_ = 0; unsigned spins = 0; bool give_up = false;
                for (;;) {
                    w_ = __hip_atomic_load(&sy->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(w_ >> 32) == slice_tag(epoch, win)) break;
                    if (++spins > SM_SPIN_LIMIT) { give_up = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (give_up && lane == 0) { __hip_atomic_store(&sy->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (info) atomicOr(info, 4); }
                if (lane == 0) bcast[win & 1] = give_up ? ~0ULL : w_;
            }
            __syncthreads();
            const unsigned long long word = bcast[win & 1];
            if (word == ~0ULL) break;
            const unsigned lo = (unsigned)word;
            if (lo & SLICE_SOLO_BIT) break;                           // the walk found the grid incomplete and left before touching anything: so do we
            const int k = (int)(lo & 0x3fffffffu);
            const bool final = (lo & SLICE_FINAL_BIT) != 0;
            if (k > 0) {
                double cv0[8], cv1[8];
                sm_flush_load_c(Cp, lane, cv0, cv1);
                for (int tile = (int)blockIdx.x - 1; tile < n_tiles; tile += 2 * F) {      // two tiles at a time
                    const int tile2 = tile + F;                                            // (workgroup-uniform)
                    SmFlushOps o1, o2;
                    sm_flush_load(o1, G, GT, Ap, Bp, (tile % tiles_per_dim) * 32 + (wave & 1) * 16, (tile / tiles_per_dim) * 32 + (wave >> 1) * 16, n, lane);
                    if (tile2 < n_tiles)
                        sm_flush_load(o2, G, GT, Ap, Bp, (tile2 % tiles_per_dim) * 32 + (wave & 1) * 16, (tile2 / tiles_per_dim) * 32 + (wave >> 1) * 16, n, lane);
                    if (o1.a0 < n && o1.b0 < n) sm_flush_apply(o1, cv0, cv1, G, GT, n, k, lane);
                    if (tile2 < n_tiles && o2.a0 < n && o2.b0 < n) sm_flush_apply(o2, cv0, cv1, G, GT, n, k, lane);
                }
            }
            if (final) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains before the arrival is signalled
            __syncthreads();
            // arrival: this workgroup's own word (a read-modify-write on one shared counter serialises in L2: 162 of them per window at n = 576)
            if (t == 0) __hip_atomic_store(&sy->arrive[blockIdx.x - 1], slice_tag(epoch, win), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // ================= walk role =================
        SmShared sh;
        sm_shared_init(sh, smem, n);
        const double* tab_g = reinterpret_cast<const double*>(d.tabs + chain);
        const long slice_off = (long)l * n;
        int8_t* fields_g = d.fields + (long)chain * d.f_stride + slice_off;
        for (int tt = t; tt < n; tt += 256) {
            // proposal position tt of this slice: site, old / new field value, ratio tables (source/model.cpp:99-122); none of it
            // depends on G or on earlier flips of the slice (each site is visited once)
            const long off = (long)chain * d.rs_stride + slice_off + tt;
            const int i = d.perm[off];
            const int kp = d.kprop[off];
            const int old_f = fields_g[i];
            sh.site[tt] = i; sh.posof[i] = tt; sh.newf[tt] = (signed char)c_proposal_sm[old_f][kp];
            sh.rbv[tt] = tab_g[old_f * 3 + kp]; sh.dlt[tt] = tab_g[12 + old_f * 3 + kp]; sh.ur[tt] = d.u[off];
            sh.diag0[tt] = G[tt + (long)n * tt];                         // first window: G was written by the previous kernel
        }
        if (t < 32) sh.tl[t] = tab_g[t];
        // census BEFORE the first window (nothing has been modified yet): has every flush workgroup checked in for this launch?  If some
        // have not after a bounded wait, the walk publishes the abort flag and leaves: the slice stays exactly as it was (fields, exp(V)
        // tables and G untouched), info |= 8 tells the host, the engine takes the kernel pairs from then on.  Once all have checked in,
        // every later hand-off completes in bounded time (resident workgroups always make progress).
        // latch: once a slice of this engine has been abandoned (info & 8), no later launch of the same sweep may update either -- the sweep
        // would otherwise contain a slice that proposed nothing between slices that did, a trajectory that belongs to no Markov chain the
        // caller asked for.  Later launches publish the abort word at once (no census wait) and leave; the host reports the first
        // abandoned slice (info[1]) and what state the engine is in (engine.hip: sync_and_check).
        const bool latched = info && (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 8);
        if (wave == 0) {
            unsigned spins = 0; bool all_in = false;
            for (;;) {
                if (latched) break;
                bool in = true;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int f = lane + 64 * q;
                    const unsigned a = f < F ? __hip_atomic_load(&sy->arrive[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : slice_tag(epoch, 0);
                    in = in && a == slice_tag(epoch, 0);
                }
                if (__all(in)) { all_in = true; break; }
                if (++spins > SM_CENSUS_SPINS) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (!all_in && lane == 0) {
                __hip_atomic_store(&sy->seq, ((unsigned long long)slice_tag(epoch, 1) << 32) | SLICE_SOLO_BIT | SLICE_FINAL_BIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&sy->solo_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (info) { atomicOr(info, 8); atomicCAS(info + 1, 0, l + 1); }      // info[1]: first abandoned slice + 1
                d.acc_out[(long)chain * d.acc_stride + acc_slot] = 0;               // nothing was proposed in this slice
            }
        }
        __syncthreads();                                              // (orders the prologue's LDS writes as well)
        // every wave re-derives the verdict from the word wave 0 may have published (LDS would need one more barrier)
        {
            const unsigned long long w0 = __hip_atomic_load(&sy->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w0 >> 32) == slice_tag(epoch, 1) && ((unsigned)w0 & SLICE_SOLO_BIT)) return;
        }
        int pos = 0, total_acc = 0;
        bool broken = false;
        SMX(SmProf prof; unsigned long long tk0, t_hand = 0; SMT(tk0))
        for (unsigned win = 1;; ++win) {
            const int k = sm_walk_window<NS>(sh, pos, n, t, G, GT, d, slice_off, chain, fields_g, Ap, Bp, Cp, win > 1 SM_PROF_PASS);
            total_acc += k;
            const bool final = pos >= n;
            SMX(unsigned long long th0; SMT(th0))
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // panel stores of every wave have left the CU
            __syncthreads();
            if (t == 0) __hip_atomic_store(&sy->seq, ((unsigned long long)slice_tag(epoch, win) << 32) | (final ? SLICE_FINAL_BIT : 0u) | (unsigned)k,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (final) break;
            // wait until every flush workgroup has absorbed this window, then refresh the diagonal from the flushed G
            if (wave == 0) {
                unsigned spins = 0;
                for (;;) {                                            // lane f, f + 64, f + 128 <-> flush workgroups (F <= 192): coalesced loads, no atomics
                    bool all_in = true;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int f = lane + 64 * q;
                        const unsigned a = f < F ? __hip_atomic_load(&sy->arrive[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : slice_tag(epoch, win);
                        all_in = all_in && a == slice_tag(epoch, win);
                    }
                    if (__all(all_in)) break;
                    if (++spins > SM_SPIN_LIMIT) { broken = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (broken && lane == 0) { __hip_atomic_store(&sy->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (info) atomicOr(info, 4); }
            }
            __syncthreads();                                          // G changed: the next window reloads the diagonal and restarts its prefetch
            SMX({ unsigned long long th1; SMT(th1) t_hand += th1 - th0; })
        }
        if (t == 0) d.acc_out[(long)chain * d.acc_stride + acc_slot] = total_acc;
        SMX(if (t == 0) { unsigned long long tk1; SMT(tk1)
            printf("sm slice l=%d: total %llu cyc | windows (walk incl. passes+accepts) %llu | %d passes %llu | %d accepts %llu (pref %llu, bcast + axpy %llu, pick + panel writes %llu, barrier %llu) | dump %llu | hand-offs %llu\n",
                   l, tk1 - tk0, prof.t_first, prof.n_pass, prof.t_pass, prof.n_acc, prof.t_acc, prof.a1, prof.a2, prof.a3, prof.a4, prof.t_dump, t_hand); })
    }
}

int update_sm_init_device() {
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(slice_sm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(slice_sm_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(slice_sm_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(slice_sm_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

// flush workgroups of the persistent slice kernels: one per 32 x 32 tile while there are at most 192 tiles, else every workgroup
// owns ceil(tiles / 192) tiles (n = 576: 324 tiles, two per workgroup, 162 workgroups)
int slice_flush_workgroups(int n) {
    const int tiles = (n + 31) / 32, T = tiles * tiles;
    if (T <= 192) return T;
    const int per = (T + 191) / 192;
    return (T + per - 1) / per;
}

// the whole local update of time slice l in one launch (caller: launch_update_slice, which has checked GT and the CU reservation)
int launch_update_slice_sm(const UpdateDesc& d, int l, int acc_slot, int n_chains, hipStream_t s) {
    const int n = d.n;
    if (n > 1024 || !d.Cpanel || !d.slice_sync || !d.GT.p) { set_error("sub-matrix slice kernel: n <= 1024, panel / sync / transposed workspaces required"); return -1; }
    const int tiles = (n + 31) / 32;
    const dim3 grid(1 + slice_flush_workgroups(n), n_chains), block(256);
    const size_t lds = sm_lds_bytes(n);
#define SM_LAUNCH(NS) hipLaunchKernelGGL((slice_sm_kernel<NS>), grid, block, lds, s, d, reinterpret_cast<SliceSync*>(d.slice_sync), l, acc_slot, tiles, d.info)
    if (n <= 256) SM_LAUNCH(1); else if (n <= 512) SM_LAUNCH(2); else if (n <= 768) SM_LAUNCH(3); else SM_LAUNCH(4);
#undef SM_LAUNCH
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
