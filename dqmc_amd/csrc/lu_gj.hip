// lu_gj.hip -- X = A^-1 B for n <= 1024 (the arma::solve(M, RHS) of
// stablelinalg::inv_I_plus_ldr_mul_ldr / inv_invldr_plus_ldr / inv_I_plus_ldr,
// source/stablelinalg.cpp:122-125,153-155,184-186) as a blocked Gauss-Jordan
// elimination with partial pivoting, two launches per panel of 32 columns and
// no substitution phase at all:
//
//   gj_panel_kernel   (n <= 64; gj_panel_mw_kernel<NW, RPL> above that: NW waves x RPL rows per lane, one LDS barrier per step, see there)
//                     ONE wave per chain.  Lane owns rows lane, lane+64, ... of the panel in
//                     registers and runs dgetf2 on the live rows (pivot search = per-lane max +
//                     32-bit DPP wave maxima + ballot, pivot row broadcast by v_readlane into
//                     SGPRs: no LDS, no barrier, no memory access in the step loop).  A is NOT
//                     modified: the kernel emits the pivot rows S of the panel, log|det| of the
//                     pivot block P11 = L11 U11, the block itself and the inverses of its four
//                     16 x 16 diagonal triangles (one column per lane).
//   gj_update_kernel  whole chip, fp64 MFMA.  Every wave forms, for its 16 columns,
//                     U12 = U11^-1 (L11^-1 A[S, cols]) by block substitution on the matrix cores (the pivot rows
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
#include <cstdlib>
#include <utility>

namespace dq {

namespace {
constexpr int GJ_NB = 32;
using d4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One elimination step of the panel, J a compile-time constant so that every register index is static (a rolled
// step loop would put the panel into scratch).  No LDS and no memory traffic: the pivot search is a per-lane max
// over the lane's rows, two 32-bit DPP wave maxima (high word of |a|, then low word among the lanes that tie) and a
// ballot; the pivot row reaches the other lanes through v_readlane into SGPRs, which the FMAs take as operands.
template <int NR, int J>
__device__ __forceinline__ void gj_step(double (&a)[NR][GJ_NB], bool (&live)[NR], int (&mypos)[NR], int& myperm, bool& singular, int lane, int k0) {
    unsigned long long best = 0ULL; int bq = -1;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const unsigned long long kq = live[q] ? ((unsigned long long)__double_as_longlong(fabs(a[q][J])) | 1ULL) : 0ULL;   // live rows have a nonzero key
        if (kq > best) { best = kq; bq = q; }             // strict: the lowest row of the lane wins a tie
    }
    const unsigned hi = (unsigned)(best >> 32), lo = (unsigned)best;
    const unsigned m1 = wave_max_u32(hi);
    const unsigned m2 = wave_max_u32(hi == m1 ? lo : 0u);
    const unsigned long long winners = __ballot(hi == m1 && lo == m2 && bq >= 0);
    const int pl = winners ? (int)__builtin_ctzll(winners) : 0;             // winners == 0 only when no live row is left (n < k0 + J + 1: excluded by the caller)
    const int pq = __builtin_amdgcn_readlane(bq, pl);
    if (lane == J) myperm = pl + 64 * pq;
    double prow[GJ_NB - J];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        if (q == pq) {                                    // wave-uniform
#pragma unroll
            for (int c = J; c < GJ_NB; ++c) prow[c - J] = readlane_f64(a[q][c], pl);
        }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q)
        if (q == pq && lane == pl) { live[q] = false; mypos[q] = k0 + J; }
    const double piv = prow[0];
    singular = singular || !(fabs(piv) > 0.0);
    // reciprocal pivot (dgetf2 scales by the reciprocal as well): v_rcp_f64 + two Newton steps, no div_scale / div_fixup
    double r = __builtin_amdgcn_rcp(piv);
    r = fma(fma(-piv, r, 1.0), r, r);
    r = fma(fma(-piv, r, 1.0), r, r);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        if (live[q]) {
            const double l = a[q][J] * r;
            a[q][J] = l;
#pragma unroll
            for (int c = J + 1; c < GJ_NB; ++c) a[q][c] = fma(-l, prow[c - J], a[q][c]);
        }
    }
}
template <int NR, int J>
struct GjSteps {
    static __device__ __forceinline__ void run(double (&a)[NR][GJ_NB], bool (&live)[NR], int (&mypos)[NR], int& myperm, bool& singular, int lane, int nbw, int k0) {
        if (J < nbw) gj_step<NR, J>(a, live, mypos, myperm, singular, lane, k0);      // wave-uniform
        GjSteps<NR, J + 1>::run(a, live, mypos, myperm, singular, lane, nbw, k0);
    }
};
template <int NR>
struct GjSteps<NR, GJ_NB> {
    static __device__ __forceinline__ void run(double (&)[NR][GJ_NB], bool (&)[NR], int (&)[NR], int&, bool&, int, int, int) {}
};
}  // namespace

// tinv: per chain 2048 doubles: [0, 1024) the pivot block LU[j][c] row-major in pivot order (L11 strictly below, U11 on and
// above the diagonal), then the four 16 x 16 triangular inverses, column-major: L11[0:16,0:16]^-1, L11[16:32,16:32]^-1,
// U11[0:16,0:16]^-1, U11[16:32,16:32]^-1.  Rows / columns >= nbw are the identity.
template <int NR>
__global__ __launch_bounds__(64) void gj_panel_kernel(CMat Am, int* rowpos_p, long rowpos_stride, int* perm_p, long perm_stride, double* tinv_p,
                                                      double* logabsdet, int accumulate, int* info, int n, int k0) {
    __shared__ __attribute__((aligned(16))) double LU[GJ_NB][GJ_NB];
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
    int myperm = 0; bool singular = false;

    GjSteps<NR, 0>::run(a, live, mypos, myperm, singular, lane, nbw, k0);

    lds_wait();                                          // the identity fill is in LDS before the pivot rows overwrite theirs
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int r = lane + 64 * q;
        if (r < n) { if (k0 == 0) rowpos[r] = mypos[q]; else if (mypos[q] >= 0) rowpos[r] = mypos[q]; }
        if (mypos[q] >= 0) {                             // a pivot row: untouched since its step, L11[j][:j] | U11[j][j:]
            double* dst = &LU[mypos[q] - k0][0];
#pragma unroll
            for (int c = 0; c < GJ_NB; c += 2) *reinterpret_cast<double2*>(dst + c) = make_double2(a[q][c], a[q][c + 1]);
        }
    }
    if (lane < nbw) perm[k0 + lane] = myperm;
    lds_wait();
#pragma unroll
    for (int e = 0; e < GJ_NB * GJ_NB / 64; ++e) tinv[lane + 64 * e] = (&LU[0][0])[lane + 64 * e];
    {                                                    // log|det P11| and singularity check
        const double pv = lane < nbw ? fabs(LU[lane][lane]) : 1.0;
        const double ls = wave_sum(log(pv));
        if (lane == 0) {
            if (logabsdet) logabsdet[chain] = ((accumulate || k0 > 0) ? logabsdet[chain] : 0.0) + ls;
            if (info && (singular || !(ls == ls))) atomicOr(info, 1);
        }
    }
    // The four 16 x 16 triangular inverses, one column per lane (lane = 16 * block + column): unit-lower blocks by forward
    // substitution, upper blocks by the same recurrence on the index-reversed block.
    {
        const int blk = lane >> 4, c = lane & 15;
        const bool up = blk >= 2;
        const int base = (blk & 1) * 16;
        const double* T0 = &LU[0][0] + (up ? (base + 15) * 33 : base * 33);
        const int sg = up ? -1 : 1;
        const int cc = up ? 15 - c : c;                  // position of the unit vector's 1 in recurrence order
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double s0 = (j == cc) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
            for (int m = 0; m < j; ++m) {
                const double t = T0[sg * (j * 32 + m)];
                if (m & 1) s1 = fma(-t, x[m], s1); else s0 = fma(-t, x[m], s0);
            }
            const double s = s0 + s1;
            x[j] = up ? s / T0[sg * (j * 33)] : s;
        }
        double* out = tinv + GJ_NB * GJ_NB + 256 * blk + 16 * c;
#pragma unroll
        for (int j = 0; j < 16; ++j) out[up ? 15 - j : j] = x[j];
    }
}

// After a multi-wave panel, one wave each: log|det P11| with the singularity flag, and the four 16 x 16 triangular inverses of the pivot block
// (4 us when wave 0 did both, each upper-block step ending in an fp64 division: the sixteen reciprocals are formed once, in parallel).
__device__ __forceinline__ void gj_panel_logdet(double (&LU)[GJ_NB][GJ_NB], double* logabsdet, int accumulate, int* info, int chain, int lane, int nbw, int k0, int s_sing) {
    const double pv = lane < nbw ? fabs(LU[lane][lane]) : 1.0;
    const double ls = wave_sum(log(pv));
    if (lane == 0) {
        if (logabsdet) logabsdet[chain] = ((accumulate || k0 > 0) ? logabsdet[chain] : 0.0) + ls;
        if (info && (s_sing || !(ls == ls))) atomicOr(info, 1);
    }
}
template <int... Js>
__device__ __forceinline__ void gj_tri_steps(double (&x)[16], const double* T0, int sg, int cc, bool up, bool hi, double rd, std::integer_sequence<int, Js...>) {
    auto step = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double s0 = (j == cc) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
        for (int m = 0; m < j; ++m) {
            const double tt = T0[sg * (j * 32 + m)];
            if (m & 1) s1 = fma(-tt, x[m], s1); else s0 = fma(-tt, x[m], s0);
        }
        const double s = s0 + s1;
        // 1 / U[base + 15 - j][base + 15 - j] sits in lane base + 15 - j (base = 0 or 16)
        const double r0 = readlane_f64(rd, 15 - j), r1 = readlane_f64(rd, 31 - j);
        x[j] = up ? s * (hi ? r1 : r0) : s;
    };
    (step(std::integral_constant<int, Js>{}), ...);
}
__device__ __forceinline__ void gj_panel_tri_inverses(double (&LU)[GJ_NB][GJ_NB], double* tinv, int lane) {
    // one column per lane (lane = 16 * block + column): unit-lower blocks by forward substitution, upper blocks by the same recurrence on
    // the index-reversed block
    const int blk = lane >> 4, c = lane & 15;
    const bool up = blk >= 2;
    const int base = (blk & 1) * 16;
    const double* T0 = &LU[0][0] + (up ? (base + 15) * 33 : base * 33);
    const int sg = up ? -1 : 1;
    const int cc = up ? 15 - c : c;
    const double rd = 1.0 / LU[lane & 31][lane & 31];
    double x[16];
    gj_tri_steps(x, T0, sg, cc, up, (blk & 1) != 0, rd, std::make_integer_sequence<int, 16>{});
    double* out = tinv + GJ_NB * GJ_NB + 256 * blk + 16 * c;
#pragma unroll
    for (int j = 0; j < 16; ++j) out[up ? 15 - j : j] = x[j];
}

#ifdef DQ_GJ_STAMPS
// diagnostic build only (scripts/gj_stamps.py): where a step's time goes; s_memtime counts 100 MHz ticks
#define GST(i) { unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); gst[i] += t_ - gprev; gprev = t_; }
#define GST_ARGS , unsigned long long (&gst)[8], unsigned long long& gprev
#define GST_PASS , gst, gprev
#else
#define GST(i)
#define GST_ARGS
#define GST_PASS
#endif
// ---- multi-wave panel: one row per lane, W = ceil(n / 64) waves (one per SIMD) ------------------------------------------------
// The single-wave kernel above is bound by instruction issue (one wave, ~12 k instructions).  Here every wave owns 64 rows, so a
// step costs a quarter of the FMAs per wave, at the price of one workgroup barrier per step: each wave finds its own best row,
// and its candidate lane publishes {key, lane} and the row's live part (columns >= J) into the wave's LDS slot BEFORE the global
// winner is known; after the barrier every wave reads the W keys, picks the winner (largest |a|, lowest wave on ties) and reads the
// pivot row from the winner's slot.  Slots are double-buffered on the parity of J (a wave can be at most one step ahead).
template <int J, int NW, int RPL>
__device__ __forceinline__ void gj_step_mw(double (&a)[RPL][GJ_NB], unsigned& live, int (&mypos)[RPL], int& myperm, bool& singular, int lane, int wave, int k0,
                                           double (*slot_row)[NW][GJ_NB + 2], unsigned long long (*slot_key)[(NW + 1) & ~1] GST_ARGS) {
    constexpr int PAR = J & 1;
    constexpr int C0 = J & ~1;
    GST(0)
    // this lane's best row: strict comparisons keep the lowest slot (= the lowest row) on a tie
    // (the values pass through an empty asm: a select between a[0][J], a[1][J], ... is otherwise turned into a load at a[bs][J], and a
    // dynamically indexed array lives in scratch)
    double aj = a[0][J]; int bs = 0;
    if (RPL > 1) asm volatile("" : "+v"(aj));
    unsigned long long key = (live & 1u) ? ((unsigned long long)__double_as_longlong(fabs(aj)) | 1ULL) : 0ULL;
#pragma unroll
    for (int q = 1; q < RPL; ++q) {
        double v = a[q][J];
        asm volatile("" : "+v"(v));
        const unsigned long long kq = ((live >> q) & 1u) ? ((unsigned long long)__double_as_longlong(fabs(v)) | 1ULL) : 0ULL;
        if (kq > key) { key = kq; aj = v; bs = q; }
    }
    // the reciprocal of the lane's own candidate element, off the critical path (dgetf2 scales by the reciprocal pivot as
    // well): v_rcp_f64 + two Newton steps; the winner's value travels with its row
    double rme = __builtin_amdgcn_rcp(aj);
    rme = fma(fma(-aj, rme, 1.0), rme, rme);
    rme = fma(fma(-aj, rme, 1.0), rme, rme);
    const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
    const unsigned m1 = wave_max_u32(hi);
    const unsigned m2 = wave_max_u32(hi == m1 ? lo : 0u);
    const unsigned long long winners = __ballot(hi == m1 && lo == m2 && key != 0ULL);
    const int cl = winners ? (int)__builtin_ctzll(winners) : 0;
    GST(1)
    if (lane == cl) {                                     // this wave's candidate
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            if (bs == q) {                                // one of the RPL store sequences runs (the row index is a register name)
#pragma unroll
                for (int c = C0; c < GJ_NB; c += 2) *reinterpret_cast<double2*>(&slot_row[PAR][wave][c]) = make_double2(a[q][c], a[q][c + 1]);
            }
        }
        slot_row[PAR][wave][GJ_NB] = rme;
        // key: |a| with its low 9 bits replaced by {non-empty marker, slot, lane}; waves compare bits 9 and up (43 mantissa bits)
        slot_key[PAR][wave] = winners ? ((((unsigned long long)m1 << 32) | m2) & ~0x1FFULL) | (unsigned long long)(cl | (bs << 6)) | 0x100ULL : 0ULL;
    }
    GST(2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    GST(3)
    // all keys in ONE LDS round trip (slots of absent waves stay 0 = "no candidate"): reading them under `w < nw` branches
    // made each key its own dependent round trip, ~450 clk per step
    constexpr int NWK = (NW + 1) & ~1;
    ulonglong2 kp[NWK / 2];
#pragma unroll
    for (int q = 0; q < NWK / 2; ++q) kp[q] = *reinterpret_cast<const ulonglong2*>(&slot_key[PAR][2 * q]);
    unsigned long long best = kp[0].x; int pw = 0;
    if ((kp[0].y >> 9) > (best >> 9)) { best = kp[0].y; pw = 1; }
#pragma unroll
    for (int q = 1; q < NWK / 2; ++q) {                   // strict: the lowest wave wins a tie
        if ((kp[q].x >> 9) > (best >> 9)) { best = kp[q].x; pw = 2 * q; }
        if ((kp[q].y >> 9) > (best >> 9)) { best = kp[q].y; pw = 2 * q + 1; }
    }
    const int pl = (int)(best & 0x3FULL), ps = (int)((best >> 6) & 3ULL);
    GST(4)
    double prow[GJ_NB - C0];
#pragma unroll
    for (int c = C0; c < GJ_NB; c += 2) { const double2 v = *reinterpret_cast<const double2*>(&slot_row[PAR][pw][c]); prow[c - C0] = v.x; prow[c + 1 - C0] = v.y; }
    const double r = slot_row[PAR][pw][GJ_NB];
    GST(5)
    if (lane == J && wave == 0) myperm = pl + 64 * pw + 64 * NW * ps;
    const bool mine = wave == pw && lane == pl;
    singular = singular || !(fabs(prow[J - C0]) > 0.0);
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        if (mine && ps == q) { live &= ~(1u << q); mypos[q] = k0 + J; }
        // branch-free: dead rows (earlier pivots, rows >= n) run the FMAs with l = 0, so the next step's pivot search can be
        // scheduled into them
        const bool lv = (live >> q) & 1u;
        const double l = lv ? a[q][J] * r : 0.0;
        a[q][J] = lv ? l : a[q][J];
#pragma unroll
        for (int c = J + 1; c < GJ_NB; ++c) a[q][c] = fma(-l, prow[c - C0], a[q][c]);
    }
    GST(6)
}
template <int J, int NW, int RPL>
struct GjStepsMW {
    static __device__ __forceinline__ void run(double (&a)[RPL][GJ_NB], unsigned& live, int (&mypos)[RPL], int& myperm, bool& singular, int lane, int wave, int nbw, int k0,
                                               double (*slot_row)[NW][GJ_NB + 2], unsigned long long (*slot_key)[(NW + 1) & ~1] GST_ARGS) {
        if (J < nbw) gj_step_mw<J, NW, RPL>(a, live, mypos, myperm, singular, lane, wave, k0, slot_row, slot_key GST_PASS);      // workgroup-uniform
        GjStepsMW<J + 1, NW, RPL>::run(a, live, mypos, myperm, singular, lane, wave, nbw, k0, slot_row, slot_key GST_PASS);
    }
};
template <int NW, int RPL>
struct GjStepsMW<GJ_NB, NW, RPL> {
    static __device__ __forceinline__ void run(double (&)[RPL][GJ_NB], unsigned&, int (&)[RPL], int&, bool&, int, int, int, int, double (*)[NW][GJ_NB + 2],
                                               unsigned long long (*)[(NW + 1) & ~1] GST_ARGS) {}
};
// NW waves (the most the workgroup may have), RPL matrix rows per lane: row t + 64 NW q is slot q of thread t.  One wave per SIMD up to n = 768
// (<4, 1> n <= 256, <4, 2> <= 512, <3, 3> <= 576, <4, 3> <= 768), two beyond (<8, 2>): with one row per lane n = 576 took 9 waves, three on
// a SIMD, and a step cost what the three issue one after the other plus nine single-lane row publishes instead of three.
template <int NW, int RPL>
__global__ __launch_bounds__(64 * NW) void gj_panel_mw_kernel(CMat Am, int* rowpos_p, long rowpos_stride, int* perm_p, long perm_stride, double* tinv_p,
                                                          double* logabsdet, int accumulate, int* info, int n, int k0) {
    constexpr int NWK = (NW + 1) & ~1;
    __shared__ __attribute__((aligned(16))) double LU[GJ_NB][GJ_NB];
    __shared__ __attribute__((aligned(16))) double slot_row[2][NW][GJ_NB + 2];     // row | reciprocal of its pivot element
    __shared__ __attribute__((aligned(16))) unsigned long long slot_key[2][NWK];
    __shared__ int s_sing;
    static_assert(RPL <= 4, "two key bits name the slot");
    const int chain = blockIdx.y;
    const double* __restrict__ A = Am.at(chain);
    int* rowpos = rowpos_p + (long)chain * rowpos_stride;
    int* perm = perm_p + (long)chain * perm_stride;
    double* tinv = tinv_p + (long)chain * 2 * GJ_NB * GJ_NB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef DQ_GJ_STAMPS
    unsigned long long gentry;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gentry) :: "memory");
#endif
    const int nbw = min(GJ_NB, n - k0);
    unsigned live = 0u;
    int mypos[RPL], myperm = 0; bool singular = false;
    double a[RPL][GJ_NB];
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int r = t + 64 * NW * q;
        const bool lv = r < n && (k0 == 0 ? true : rowpos[r < n ? r : 0] < 0);
        live |= lv ? (1u << q) : 0u;
        mypos[q] = -1;
#pragma unroll
        for (int c = 0; c < GJ_NB; ++c) a[q][c] = (lv && c < nbw) ? A[r + (long)n * (k0 + c)] : 0.0;
    }
    for (int e = t; e < GJ_NB * GJ_NB; e += blockDim.x) (&LU[0][0])[e] = ((e >> 5) == (e & 31)) ? 1.0 : 0.0;   // identity padding for nbw < 32
    if (t == 0) s_sing = 0;
    if (t < 2 * NWK) (&slot_key[0][0])[t] = 0ULL;         // waves that do not exist never publish: their keys stay "no candidate"
    __syncthreads();

#ifdef DQ_GJ_STAMPS
    unsigned long long gst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gprev, gstart;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gprev) :: "memory"); gstart = gprev;
#endif
    GjStepsMW<0, NW, RPL>::run(a, live, mypos, myperm, singular, lane, wave, nbw, k0, slot_row, slot_key GST_PASS);
#ifdef DQ_GJ_STAMPS
    if (t == 0 && k0 == 0 && blockIdx.y == 0)
        printf("gj panel k0=0 n=%d, %d waves x %d rows per lane, clocks over %d steps (each segment includes one stamp, ~130): (loop) %llu | search %llu | publish %llu | barrier %llu | keys %llu | row %llu | fma %llu | total %llu\n",
               n, (int)(blockDim.x >> 6), RPL, nbw, gst[0], gst[1], gst[2], gst[3], gst[4], gst[5], gst[6], gprev - gstart);
#endif

    __syncthreads();                                     // identity fill done everywhere before the pivot rows overwrite theirs
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int r = t + 64 * NW * q;
        if (r < n) { if (k0 == 0) rowpos[r] = mypos[q]; else if (mypos[q] >= 0) rowpos[r] = mypos[q]; }
        if (mypos[q] >= 0) {
            double* dst = &LU[mypos[q] - k0][0];
#pragma unroll
            for (int c = 0; c < GJ_NB; c += 2) *reinterpret_cast<double2*>(dst + c) = make_double2(a[q][c], a[q][c + 1]);
        }
    }
    if (wave == 0 && lane < nbw) perm[k0 + lane] = myperm;
    if (singular && lane == 0) s_sing = 1;
    __syncthreads();
    for (int e = t; e < GJ_NB * GJ_NB; e += blockDim.x) tinv[e] = (&LU[0][0])[e];
#ifdef DQ_GJ_STAMPS
    unsigned long long gmid;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gmid) :: "memory");
#endif
    if (wave == 0) gj_panel_tri_inverses(LU, tinv, lane);
    else if (wave == 1) gj_panel_logdet(LU, logabsdet, accumulate, info, chain, lane, nbw, k0, s_sing);
#ifdef DQ_GJ_STAMPS
    unsigned long long gend;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gend) :: "memory");
    if (t == 0 && k0 == 0 && blockIdx.y == 0) printf("   prologue %llu | steps %llu | LU hand-over (+ printf) %llu | epilogue %llu\n", gstart - gentry, gprev - gstart, gmid - gprev, gend - gmid);
#endif
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
    const double* __restrict__ LUg = tinv_p + (long)chain * 2 * GJ_NB * GJ_NB;      // pivot block, row-major
    const double* __restrict__ Tinv = LUg + GJ_NB * GJ_NB;                          // 4 x (16 x 16) triangular inverses, column-major
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
    // ---- triangular operands (A operand of the MFMA: row r16, k = 4 s + kk) ----
    double lai[4], lbi[4], uai[4], ubi[4], cl[4], cu[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k = 4 * s + kk;
        lai[s] = Tinv[0 * 256 + r16 + 16 * k]; lbi[s] = Tinv[1 * 256 + r16 + 16 * k];
        uai[s] = Tinv[2 * 256 + r16 + 16 * k]; ubi[s] = Tinv[3 * 256 + r16 + 16 * k];
        cl[s] = -LUg[(16 + r16) * 32 + k];               // -L11[16 + r16][k]
        cu[s] = -LUg[r16 * 32 + 16 + k];                 // -U11[r16][16 + k]
    }
    // ---- T1 = L11^-1 A12 by block forward substitution.  The B-operand chunking k = 4 s + kk is the D layout
    //      (row kk + 4 reg), so a12[1] seeds the accumulator of the Schur step directly. ----
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    d4 t1[2], u[2];
    t1[0] = zero4;
#pragma unroll
    for (int s = 0; s < 4; ++s) t1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(lai[s], a12[0][s], t1[0], 0, 0, 0);
    d4 w = {a12[1][0], a12[1][1], a12[1][2], a12[1][3]};
#pragma unroll
    for (int s = 0; s < 4; ++s) w = __builtin_amdgcn_mfma_f64_16x16x4f64(cl[s], t1[0][s], w, 0, 0, 0);
    t1[1] = zero4;
#pragma unroll
    for (int s = 0; s < 4; ++s) t1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(lbi[s], w[s], t1[1], 0, 0, 0);
    // ---- U12 = U11^-1 T1 by block back substitution ----
    u[1] = zero4;
#pragma unroll
    for (int s = 0; s < 4; ++s) u[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ubi[s], t1[1][s], u[1], 0, 0, 0);
    w = t1[0];
#pragma unroll
    for (int s = 0; s < 4; ++s) w = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[s], u[1][s], w, 0, 0, 0);
    u[0] = zero4;
#pragma unroll
    for (int s = 0; s < 4; ++s) u[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(uai[s], w[s], u[0], 0, 0, 0);
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
    if (n > 1024) { set_error("gj_solve supports n <= 1024"); return -1; }
    for (int k0 = 0; k0 < n; k0 += GJ_NB) {
        const dim3 pg(1, n_chains);
#define DQ_GJ_PANEL(NW, RPL) hipLaunchKernelGGL((gj_panel_mw_kernel<NW, RPL>), pg, dim3(RPL == 1 ? 64 * ((n + 63) / 64) : 64 * NW), 0, s, CMat(A), rowpos, rowpos_stride, perm, perm_stride, tinv, logabsdet, accumulate_logdet, info, n, k0)
        if (n > 768) DQ_GJ_PANEL(8, 2);
        else if (n > 576) DQ_GJ_PANEL(4, 3);
        else if (n > 512) DQ_GJ_PANEL(3, 3);
        else if (n > 256) DQ_GJ_PANEL(4, 2);
        else if (n > 64) DQ_GJ_PANEL(4, 1);
        else hipLaunchKernelGGL((gj_panel_kernel<1>), pg, dim3(64), 0, s, CMat(A), rowpos, rowpos_stride, perm, perm_stride, tinv, logabsdet, accumulate_logdet, info, n, k0);      // n <= 64: one wave holds every row
        const int nbw = n - k0 < GJ_NB ? n - k0 : GJ_NB;
        const int nA = (n + 31) / 32, nS = k0 / 32;
        const int nCA = (n - k0 - nbw + 31) / 32, nCB = (n + 31) / 32;
        hipLaunchKernelGGL(gj_update_kernel, dim3((nA + nS + 1) * (nCA + nCB), n_chains), dim3(256), 0, s, A, B, SA, X, (const int*)rowpos, rowpos_stride,
                           (const int*)perm, perm_stride, (const double*)tinv, n, k0, nA, nS, nCA);
    }
#undef DQ_GJ_PANEL
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
