// qr_coop.hip -- column-pivoted Householder QR for 256 < n <= 1024 spread over P = ceil(n / 32) cooperating
// workgroups per chain (to_LDR, source/stablelinalg.cpp:35-55; LAPACK dlaqp2 semantics and the output format of
// qr.hip / qr_colown.hip: reflectors and R0 left in place in A without column swaps, jpvt / tau out).
//
// Why: beyond n = 256 the matrix no longer fits one CU (n = 576: 2.65 MB against 512 KB of registers + 160 KB of LDS), and
// the single-workgroup streaming kernel of qr.hip re-reads it from L2 every step: 23.7 ms per factorisation at n = 576,
// 59 % of a cfg-5 sweep (profiles/r02_bench_cfg5_kernel_stats_v1.csv).  Here every workgroup (256 threads, one wave per
// SIMD) owns 32 columns entirely in registers (2 columns x ceil(n/16) row slots per thread, no LDS data, no spills) and
// each of the n serial steps costs one all-to-all exchange:
//
//   step i:  every workgroup builds the Householder reflector of ITS best live column (largest partial norm)
//            speculatively and publishes {norm, column, tau, beta, v[0:n]} as data-tagged 8-byte granules
//            (tag = step + 1, relaxed agent-scope atomic stores = sc1, write-through);
//            every workgroup sweeps the P headers until the tags match, takes the record with the largest norm (lowest
//            column on ties: dgeqp3's idamax rule, exactly -- norms are compared as full doubles) -- all workgroups
//            reach the same decision without a barrier or a flag --, reads that record's payload and applies the reflector
//            to its live columns.  The winner also writes the reflector / tau / jpvt to HBM.
//
// This is recipe R2 of cdna_hip_programming.md, Guideline 16 (the data is the flag; every shared word is an 8-byte
// agent-scope access, tags never 0, buffers zeroed by a memset node before every launch, spins bounded with an abort
// word).  Records are double-buffered by step parity: a workgroup can publish step i+1 only after it has read every
// step-i header and the winner's payload, so a reader of step i never sees its slot overwritten.  Placement-independent:
// nothing relies on which XCD / CU a workgroup runs on; the P workgroups of a chain must be co-resident (the launcher
// takes this path only while P x chains fits the CU budget, and a spin that expires reports DQMC_ENUMERIC, never a hang).
// At n <= 256 the single-CU column-owner kernel (qr_colown.hip) is faster: a cross-CU round trip costs ~1.5 us here.
#include "common.h"
#include "wave.h"

namespace dq {

namespace {

constexpr int QC_COLS = 32;                 // columns per workgroup
constexpr int QC_T = 256;                   // threads per workgroup
constexpr int QC_HDR = 4;                   // header slots of a record: norm, column, tau, beta
constexpr unsigned QC_SPIN_LIMIT = 1u << 18;     // ~0.2 s of polling; one time-out raises the abort word and ends every later spin at once

using u64 = unsigned long long;

__host__ __device__ inline int qc_workgroups(int n) { return (n + QC_COLS - 1) / QC_COLS; }
__host__ __device__ inline long qc_rec_granules(int n) { return 2L * (QC_HDR + n); }      // one fp64 slot = two tagged granules

__device__ __forceinline__ void put_f64(u64* rec, int slot, double x, unsigned tag) {
    const u64 b = (u64)__double_as_longlong(x);
    __hip_atomic_store(rec + 2 * slot, ((u64)tag << 32) | (b & 0xffffffffULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 2 * slot + 1, ((u64)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns true when both granules carry `tag`
__device__ __forceinline__ bool get_f64(const u64* rec, int slot, unsigned tag, double& x) {
    const u64 lo = __hip_atomic_load(rec + 2 * slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 hi = __hip_atomic_load(rec + 2 * slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x = __longlong_as_double((long long)(((hi & 0xffffffffULL) << 32) | (lo & 0xffffffffULL)));
    return (unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag;
}
// idamax over the wave: the largest norm among the valid lanes, the lowest column among the lanes that hold it.
// Norms are non-negative, so their bit patterns order like the numbers; bit 63 marks "valid".  Returns false when no lane is valid.
__device__ __forceinline__ bool wave_argmax_norm(double nrm, int col, bool valid, int& best_col) {
    const u64 bits = valid ? ((1ULL << 63) | (u64)__double_as_longlong(nrm)) : 0ULL;
    const u64 top = wave_max_u64(bits);
    const unsigned c = (valid && bits == top) ? ~(unsigned)col : 0u;
    best_col = (int)~wave_max_u32(c);
    return top != 0ULL;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence = s_waitcnt vmcnt(0): inside the step
// loop that makes a workgroup wait for its own global stores -- the published payload, and for the step's winner the reflector
// column it records (up to n stores issued just before the barrier, a full memory latency) -- although every barrier of the
// loop only protects LDS buffers; data for other workgroups travels as tagged granules and needs no fence.  (Measured at N = 576:
// no difference, 4.5 ms per factorisation either way -- the polls that follow wait for those stores anyway; kept because it is
// the weaker and sufficient barrier.)
__device__ __forceinline__ void qc_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace

// grid = (P, chains); sync: [chains][2][P][qc_rec_granules(n)] granules (zeroed before launch); abort_word: 1 int per chain
// NB: row slots per thread (rows 16 j + rg), NB * 16 >= n
template <int NB>
__global__ __launch_bounds__(QC_T) void qrcp_coop_kernel(Mat Am, QrWork w, int n, u64* sync_p, long sync_stride, int* abort_p, int* info) {
    __shared__ double xbuf[16 * NB];        // image of my candidate column
    __shared__ double vbuf[16 * NB];        // the winning Householder vector of this step
    __shared__ double rowi[QC_COLS];        // row i of my columns (norm down-date)
    __shared__ double tails[QC_COLS];       // recomputed tail norms^2 (rare path)
    __shared__ double scal[4];              // tau, beta, scale of my candidate
    __shared__ int pposl[QC_COLS];          // pivot position of a pivoted column of mine
    const int chain = blockIdx.y, wg = blockIdx.x, P = gridDim.x;
    const long REC = qc_rec_granules(n);
    double* __restrict__ A = Am.at(chain);
    double* tau = w.tau + (long)chain * w.tau_stride;
    int* jpvt = w.jpvt + (long)chain * w.jpvt_stride;
    u64* sync = sync_p + (long)chain * sync_stride;
    int* abort_w = abort_p + chain;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int rg = lane & 15, cl = lane >> 4;
    const int lc0 = 8 * wave + 2 * cl;                 // my local columns lc0, lc0 + 1 (0..31)
    const int gc0 = QC_COLS * wg + lc0;                // global column of local column lc0
    const double tol3z = 1.0536712127723509e-08;       // sqrt(2^-53)

    double a[2][NB];
    unsigned wlive;                                    // bit q: local column q is live (same in every thread)
    { const int cnt = n - QC_COLS * wg; wlive = cnt >= 32 ? 0xFFFFFFFFu : (cnt > 0 ? ((1u << cnt) - 1u) : 0u); }
    if (t < QC_COLS) pposl[t] = 1 << 30;
    for (int r = n + t; r < 16 * NB; r += QC_T) { xbuf[r] = 0.0; vbuf[r] = 0.0; }      // padding rows: read (times a zero matrix entry) but never written

    // ---- load my columns, initial norms (every lane q < 32 of every wave keeps the norm of local column q) ----
    double nrm1 = 0.0, nrm2 = 0.0;                      // vn1 / vn2 of local column (lane & 31), valid in lanes 0..31
    {
        double n0 = 0.0, n1 = 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int r = 16 * j + rg;
            a[0][j] = (r < n && gc0 < n) ? A[r + (long)n * gc0] : 0.0;
            a[1][j] = (r < n && gc0 + 1 < n) ? A[r + (long)n * (gc0 + 1)] : 0.0;
            n0 += a[0][j] * a[0][j]; n1 += a[1][j] * a[1][j];
        }
        n0 = sqrt(row16_sum(n0)); n1 = sqrt(row16_sum(n1));
        if (rg == 0) { rowi[lc0] = n0; rowi[lc0 + 1] = n1; }     // rowi doubles as the staging buffer here
        __syncthreads();
        nrm1 = rowi[lane & 31]; nrm2 = nrm1;
        __syncthreads();
    }

    // ---- [1] my best live column (identical in all four waves: same inputs, no exchange) and its header {norm, column}.  The header
    // of step i + 1 is published as soon as the norms of step i are final -- before the reflector of step i + 1 is built and its
    // payload written -- so that the exchange of the headers (one store + one load latency across the chip) runs while every
    // workgroup builds its candidate: by the time a workgroup polls, the headers are usually there, and the step's critical path
    // is payload store -> payload load instead of header store -> header load -> payload load.
    int cbest = 0; bool have = false; int cstar = 0; double mynorm = 0.0;
    auto pick_and_announce = [&](int step) {
        have = wave_argmax_norm(nrm1, QC_COLS * wg + lane, lane < 32 && ((wlive >> lane) & 1u), cbest);
        cstar = have ? cbest - QC_COLS * wg : 0;                                        // local column 0..31
        mynorm = readlane_f64(nrm1, cstar);
        if (t == 0) {
            u64* rec = sync + ((long)(step & 1) * P + wg) * REC;
            put_f64(rec, 0, have ? mynorm : -1.0, (unsigned)step + 1u);
            put_f64(rec, 1, have ? (double)cbest : -1.0, (unsigned)step + 1u);
        }
    };
    pick_and_announce(0);
    for (int i = 0; i < n; ++i) {
        const int jb = i >> 4, nj = NB - jb;            // slot j <-> row block jb + j
        const unsigned tag = (unsigned)i + 1u;
        u64* myrec = sync + ((long)(i & 1) * P + wg) * REC;
        // ---- [2] its Householder reflector, built speculatively ----
        if (wave == (cstar >> 3)) {
            const bool mine = cl == ((cstar & 7) >> 1);
            double ss = 0.0, al = 0.0;
            // two copies behind a wave-uniform branch: `cond ? a[1][j] : a[0][j]` is turned into a runtime-indexed
            // array access, which sends the whole register image of the matrix to scratch
#define QC_PUBLISH(KC)                                                  \
            {                                                           \
                _Pragma("unroll") for (int j = 0; j < NB; ++j) {        \
                    if (j < nj) {                                       \
                        const int r = 16 * (jb + j) + rg;               \
                        const double x = a[KC][j];                      \
                        if (mine) xbuf[r] = x;                          \
                        if (r > i) ss += x * x;                         \
                    }                                                   \
                }                                                       \
                al = a[KC][0];                                          \
            }
            if (cstar & 1) QC_PUBLISH(1) else QC_PUBLISH(0)
#undef QC_PUBLISH
            ss = row16_sum(ss);
            if (mine && rg == (i & 15)) {
                double tau_l = 0.0, beta_l = al, scale_l = 0.0;
                if (ss != 0.0) {
                    beta_l = -copysign(sqrt(al * al + ss), al);
                    tau_l = (beta_l - al) / beta_l;
                    scale_l = 1.0 / (al - beta_l);
                }
                scal[0] = tau_l; scal[1] = beta_l; scal[2] = scale_l;
            }
        }
        qc_lds_barrier();
        // ---- [3] publish {norm, column, tau, beta, v} as tagged granules ----
        const int r_lo = (i / QC_T) * QC_T;                  // v is zero above row i and nobody reads rows below block jb: skip whole chunks
        for (int r = r_lo + t; r < n; r += QC_T) {
            double vmine = 0.0;
            if (have) { if (r == i) vmine = 1.0; else if (r > i) vmine = xbuf[r] * scal[2]; }
            put_f64(myrec, QC_HDR + r, vmine, tag);
        }
        if (t == 0) {
            put_f64(myrec, 2, have ? scal[0] : 0.0, tag);
            put_f64(myrec, 3, have ? scal[1] : 0.0, tag);
        }
        // ---- [4] sweep the headers (lane q < P of every wave reads record q), then the winner's payload ----
        const u64* recs = sync + (long)(i & 1) * P * REC;
        int p = 0;                                            // global pivot column
        {
            unsigned spins = 0;
            for (;;) {
                double hn = -1.0, hc = -1.0; bool ok = true;
                if (lane < P) { const bool o1 = get_f64(recs + (long)lane * REC, 0, tag, hn), o2 = get_f64(recs + (long)lane * REC, 1, tag, hc); ok = o1 && o2; }
                const bool all_ok = __all(ok);
                // bounded spin: give up, raise the chain's abort word (every later spin of every workgroup then exits at
                // once) and carry on with whatever was read -- the barrier structure stays intact, the kernel ends in
                // bounded time, the host sees info bit 1 and reports the factorisation as failed
                bool bail = false;
                if (!all_ok && (++spins > QC_SPIN_LIMIT || __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (info && lane == 0) atomicOr(info, 2);
                    bail = true;
                }
                if (all_ok || bail) {
                    int pc = 0;
                    const bool any = wave_argmax_norm(hn, (int)hc, lane < P && ok && hn >= 0.0, pc);
                    p = any ? pc : 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const int wwin = p / QC_COLS;                        // its workgroup
        const u64* wrec = recs + (long)wwin * REC;
        double tau_i = 0.0, beta = 0.0;
        {
            unsigned spins = 0;
            for (;;) {
                bool ok = get_f64(wrec, 2, tag, tau_i);
                ok = get_f64(wrec, 3, tag, beta) && ok;
                for (int r = r_lo + t; r < n; r += QC_T) { double vr; ok = get_f64(wrec, QC_HDR + r, tag, vr) && ok; vbuf[r] = vr; }
                if (__all(ok)) break;
                if (++spins > QC_SPIN_LIMIT || __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (info && lane == 0) atomicOr(info, 2);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        // ---- [5] the winner records the reflector, tau, jpvt ----
        if (wwin == wg) {
            for (int r = r_lo + t; r < n; r += QC_T) { if (r == i) A[r + (long)n * p] = beta; else if (r > i) A[r + (long)n * p] = vbuf[r]; }
            if (t == 0) { tau[i] = tau_i; jpvt[i] = p; pposl[p & 31] = i; }
            wlive &= ~(1u << (p & 31));
        }
        qc_lds_barrier();
        // ---- [6] apply H to my live columns ----
        const bool live0 = (wlive >> lc0) & 1u, live1 = (wlive >> (lc0 + 1)) & 1u;
        {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < nj) { const double vj = vbuf[16 * (jb + j) + rg]; s0 += a[0][j] * vj; s1 += a[1][j] * vj; }
            }
            s0 = row16_sum(s0) * tau_i; s1 = row16_sum(s1) * tau_i;
            if (!live0) s0 = 0.0;
            if (!live1) s1 = 0.0;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < nj) { const double vj = vbuf[16 * (jb + j) + rg]; a[0][j] -= s0 * vj; a[1][j] -= s1 * vj; }
            }
        }
        // ---- [7] norm down-date (dlaqp2): lane q < 32 of every wave owns local column q ----
        if (rg == (i & 15)) { rowi[lc0] = a[0][0]; rowi[lc0 + 1] = a[1][0]; }
        qc_lds_barrier();
        int need_col = 0;
        if (lane < 32 && ((wlive >> lane) & 1u) && nrm1 != 0.0) {
            double temp = fabs(rowi[lane]) / nrm1; temp = fmax(0.0, 1.0 - temp * temp);
            const double rr = nrm1 / nrm2;
            if (temp * rr * rr <= tol3z) need_col = 1;
            else nrm1 = nrm1 * sqrt(temp);
        }
        const unsigned needmask = (unsigned)(__ballot(need_col) & 0xFFFFFFFFULL);     // identical in all four waves
        if (needmask) {                                                               // rare: recompute the flagged norms
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < nj && 16 * (jb + j) + rg > i) { t0 += a[0][j] * a[0][j]; t1 += a[1][j] * a[1][j]; }
            }
            t0 = row16_sum(t0); t1 = row16_sum(t1);
            if (rg == 0) { tails[lc0] = t0; tails[lc0 + 1] = t1; }
            qc_lds_barrier();
            if (lane < 32 && ((needmask >> lane) & 1u)) { nrm1 = (i + 1 < n) ? sqrt(tails[lane]) : 0.0; nrm2 = nrm1; }
            qc_lds_barrier();
        }
        // ---- [1'] the norms are final: candidate and header of the next step ----
        if (i + 1 < n) pick_and_announce(i + 1);
        // ---- [8] every 16 steps (and at the end): row block jb is final -> store it as R0, shift the registers ----
        if ((i & 15) == 15 || i == n - 1) {
            const int r = 16 * jb + rg;
            if (r < n) {
                if (gc0 < n && (live0 ? r <= i : r < pposl[lc0])) A[r + (long)n * gc0] = a[0][0];
                if (gc0 + 1 < n && (live1 ? r <= i : r < pposl[lc0 + 1])) A[r + (long)n * (gc0 + 1)] = a[1][0];
            }
            if ((i & 15) == 15) {
#pragma unroll
                for (int j = 0; j < NB - 1; ++j) { a[0][j] = a[0][j + 1]; a[1][j] = a[1][j + 1]; }
            }
        }
    }
}

long qrcp_coop_sync_granules(int n) { return 2L * qc_workgroups(n) * qc_rec_granules(n); }
// how many CUs the cooperating workgroups of one launch occupy (one workgroup per CU at these register counts)
int qrcp_coop_workgroups(int n, int n_chains) { return qc_workgroups(n) * n_chains; }

int launch_qrcp_coop(Mat A, QrWork w, int n, int n_chains, hipStream_t s) {
    if (n > 1024) { set_error("cooperative QRCP supports n <= 1024"); return -1; }
    if (!w.sync || w.sync_stride < qrcp_coop_sync_granules(n) || !w.abort_words) { set_error("cooperative QRCP needs its sync workspace"); return -1; }
    DQ_HIP(hipMemsetAsync(w.sync, 0, sizeof(unsigned long long) * w.sync_stride * n_chains, s));
    DQ_HIP(hipMemsetAsync(w.abort_words, 0, sizeof(int) * n_chains, s));
    const dim3 grid(qc_workgroups(n), n_chains), block(QC_T);
#define QC_LAUNCH(NB) hipLaunchKernelGGL((qrcp_coop_kernel<NB>), grid, block, 0, s, A, w, n, (unsigned long long*)w.sync, w.sync_stride, w.abort_words, w.info)
    if (n <= 256) QC_LAUNCH(16); else if (n <= 384) QC_LAUNCH(24); else if (n <= 576) QC_LAUNCH(36); else if (n <= 768) QC_LAUNCH(48); else QC_LAUNCH(64);
#undef QC_LAUNCH
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
