// qr_coop.hip -- column-pivoted Householder QR for 128 < n <= 256 spread over
// QC_P = 8 cooperating workgroups per chain (to_LDR, source/stablelinalg.cpp:35-55; same
// LAPACK dlaqp2 semantics and output format as qr.hip / qr_onchip.hip: reflectors
// and R0 left in place in A, jpvt / tau out).
//
// Why: one CU cannot hold a 256 x 256 fp64 matrix next to its working registers
// (qr_onchip.hip: 2 of 8 columns per thread live in LDS, whose write bandwidth
// then paces the update: 1.2 ms per factorisation), and a single CU's fp64 rate
// sets a 0.1 ms floor anyway.  Here every workgroup (256 threads, one wave per
// SIMD) owns 32 columns entirely in registers (32 doubles per thread, no LDS data,
// no spills) and the 256 serial steps cost one all-to-all exchange each:
//
//   step i:  every workgroup builds the Householder reflector of ITS best live
//            column (largest partial norm) speculatively and publishes
//            {key, tau, beta, v[0:256]} as data-tagged 8-byte granules
//            (tag = step + 1, relaxed agent-scope atomic stores = sc1, write-through);
//            every workgroup sweeps all 8 records until the tags match, takes the
//            record with the largest key -- all workgroups reach the same decision
//            without a barrier or a flag -- and applies that reflector to its live
//            columns.  The winner also writes the reflector / tau / jpvt to HBM.
//
// This is recipe R2 of cdna_hip_programming.md, Guideline 16 (the data is the flag;
// every shared word is an 8-byte agent-scope access, tags never 0, buffers zeroed by a
// memset node before every launch, spins bounded with an abort word).  Records are
// double-buffered by step parity: a workgroup can publish step i+1 only after it has
// read every step-i record, so a reader of step i never sees its slot overwritten.
// Placement-independent: nothing relies on which XCD / CU a workgroup runs on; the 8
// workgroups of a chain must be co-resident (8 x chains <= resident slots; checked
// by the launcher against the CU count).
#include "common.h"
#include "wave.h"

namespace dq {

namespace {

constexpr int QC_P = 8;                     // workgroups per chain
constexpr int QC_COLS = 32;                 // columns per workgroup
constexpr int QC_T = 256;                   // threads per workgroup
constexpr int QC_REC = 8 + 2 * 256;         // granules per record: header (key, tau, beta, spare) + v[256]
constexpr unsigned QC_SPIN_LIMIT = 1u << 18;     // ~0.2 s of polling; one time-out raises the abort word and ends every later spin at once

using u64 = unsigned long long;

__device__ __forceinline__ u64 qc_key(double nrm, int c) {
    return (1ULL << 63) | ((u64)__double_as_longlong(nrm) & ~0xFFULL) | (u64)(255 - c);
}
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

}  // namespace

// grid = (QC_P, chains); sync: [chains][2][QC_P][QC_REC] granules (zeroed before launch); abort_word: 1 int per chain
__global__ __launch_bounds__(QC_T) void qrcp_coop_kernel(Mat Am, QrWork w, int n, u64* sync_p, long sync_stride, int* abort_p, int* info) {
    __shared__ double xbuf[256];            // image of my candidate column
    __shared__ double vbuf[256];            // the winning Householder vector of this step
    __shared__ double rowi[QC_COLS];        // row i of my columns (norm down-date)
    __shared__ double tails[QC_COLS];       // recomputed tail norms^2 (rare path)
    __shared__ double scal[4];              // tau, beta, scale of my candidate
    __shared__ int pposl[QC_COLS];          // pivot position of a pivoted column of mine
    const int chain = blockIdx.y, wg = blockIdx.x;
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

    double a[2][16];
    unsigned wlive;                                    // bit q: local column q is live (same in every thread)
    { const int cnt = n - QC_COLS * wg; wlive = cnt >= 32 ? 0xFFFFFFFFu : (cnt > 0 ? ((1u << cnt) - 1u) : 0u); }
    if (t < QC_COLS) pposl[t] = 256;

    // ---- load my columns, initial norms (every lane q < 32 of every wave keeps the norm of local column q) ----
    double nrm1 = 0.0, nrm2 = 0.0;                      // vn1 / vn2 of local column (lane & 31), valid in lanes 0..31
    {
        double n0 = 0.0, n1 = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
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

    for (int i = 0; i < n; ++i) {
        const int jb = i >> 4, nj = 16 - jb;            // slot j <-> row block jb + j
        const unsigned tag = (unsigned)i + 1u;
        u64* myrec = sync + ((long)(i & 1) * QC_P + wg) * QC_REC;
        // ---- [1] my best live column (identical in all four waves: same inputs, no exchange) ----
        u64 mykey = (lane < 32 && ((wlive >> lane) & 1u)) ? qc_key(nrm1, QC_COLS * wg + lane) : 0ULL;
        mykey = wave_max_u64(mykey);
        const int cstar = mykey ? (255 - (int)(mykey & 0xFFULL)) - QC_COLS * wg : 0;   // local column 0..31
        // ---- [2] its Householder reflector, built speculatively ----
        if (wave == (cstar >> 3)) {
            const bool mine = cl == ((cstar & 7) >> 1);
            double ss = 0.0, al = 0.0;
            // two copies behind a wave-uniform branch: `cond ? a[1][j] : a[0][j]` is turned into a runtime-indexed
            // array access, which sends the whole register image of the matrix to scratch
#define QC_PUBLISH(KC)                                                  \
            {                                                           \
                _Pragma("unroll") for (int j = 0; j < 16; ++j) {        \
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
        __syncthreads();
        // ---- [3] publish {key, tau, beta, v} as tagged granules ----
        double vmine = 0.0;
        if (mykey) {
            if (t == i) vmine = 1.0; else if (t > i) vmine = xbuf[t] * scal[2];
        }
        put_f64(myrec, 4 + t, vmine, tag);
        if (t == 0) {
            const u64 k = mykey;
            __hip_atomic_store(myrec + 0, ((u64)tag << 32) | (k & 0xffffffffULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(myrec + 1, ((u64)tag << 32) | (k >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            put_f64(myrec, 1, mykey ? scal[0] : 0.0, tag);
            put_f64(myrec, 2, mykey ? scal[1] : 0.0, tag);
        }
        // ---- [4] sweep the records: keys first (lanes 0..7 of every wave, one record each), then the winner's payload.
        //          (Reading all eight payloads speculatively in the same pass was tried: 64 sc1 loads per thread per poll
        //          cost more than the dependent round trip they save: 1.83 ms vs 1.26 ms per factorisation.) ----
        const u64* recs = sync + (long)(i & 1) * QC_P * QC_REC;
        u64 best = 0ULL;
        {
            unsigned spins = 0;
            for (;;) {
                u64 k = 0ULL; bool ok = true;
                if (lane < QC_P) {
                    const u64 lo = __hip_atomic_load(recs + (long)lane * QC_REC + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const u64 hi = __hip_atomic_load(recs + (long)lane * QC_REC + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag;
                    k = ((hi & 0xffffffffULL) << 32) | (lo & 0xffffffffULL);
                }
                if (__all(ok)) { best = wave_max_u64(k); break; }
                // bounded spin: give up, raise the chain's abort word (every later spin of every workgroup then exits at
                // once) and carry on with whatever was read -- the barrier structure stays intact, the kernel ends in
                // bounded time, the host sees info bit 1 and reports the factorisation as failed
                if (++spins > QC_SPIN_LIMIT || __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (info && lane == 0) atomicOr(info, 2);
                    best = wave_max_u64(k); break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const int p = 255 - (int)(best & 0xFFULL);           // global pivot column
        const int wwin = p >> 5;                             // its workgroup
        const u64* wrec = recs + (long)wwin * QC_REC;
        double tau_i = 0.0, beta = 0.0, vr = 0.0;
        {
            unsigned spins = 0;
            for (;;) {
                const bool ok_v = get_f64(wrec, 4 + t, tag, vr), ok_t = get_f64(wrec, 1, tag, tau_i), ok_b = get_f64(wrec, 2, tag, beta);
                const bool ok = ok_v && ok_t && ok_b;
                if (__all(ok)) break;
                if (++spins > QC_SPIN_LIMIT || __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (info && lane == 0) atomicOr(info, 2);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        vbuf[t] = vr;
        // ---- [5] the winner records the reflector, tau, jpvt ----
        if (wwin == wg) {
            if (t < n) { if (t == i) A[t + (long)n * p] = beta; else if (t > i) A[t + (long)n * p] = vr; }
            if (t == 0) { tau[i] = tau_i; jpvt[i] = p; pposl[p & 31] = i; }
            wlive &= ~(1u << (p & 31));
        }
        __syncthreads();
        // ---- [6] apply H to my live columns ----
        const bool live0 = (wlive >> lc0) & 1u, live1 = (wlive >> (lc0 + 1)) & 1u;
        {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < nj) { const double vj = vbuf[16 * (jb + j) + rg]; s0 += a[0][j] * vj; s1 += a[1][j] * vj; }
            }
            s0 = row16_sum(s0) * tau_i; s1 = row16_sum(s1) * tau_i;
            if (!live0) s0 = 0.0;
            if (!live1) s1 = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < nj) { const double vj = vbuf[16 * (jb + j) + rg]; a[0][j] -= s0 * vj; a[1][j] -= s1 * vj; }
            }
        }
        // ---- [7] norm down-date (dlaqp2): lane q < 32 of every wave owns local column q ----
        if (rg == (i & 15)) { rowi[lc0] = a[0][0]; rowi[lc0 + 1] = a[1][0]; }
        __syncthreads();
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
            for (int j = 0; j < 16; ++j) {
                if (j < nj && 16 * (jb + j) + rg > i) { t0 += a[0][j] * a[0][j]; t1 += a[1][j] * a[1][j]; }
            }
            t0 = row16_sum(t0); t1 = row16_sum(t1);
            if (rg == 0) { tails[lc0] = t0; tails[lc0 + 1] = t1; }
            __syncthreads();
            if (lane < 32 && ((needmask >> lane) & 1u)) { nrm1 = (i + 1 < n) ? sqrt(tails[lane]) : 0.0; nrm2 = nrm1; }
            __syncthreads();
        }
        // ---- [8] every 16 steps (and at the end): row block jb is final -> store it as R0, shift the registers ----
        if ((i & 15) == 15 || i == n - 1) {
            const int r = 16 * jb + rg;
            if (r < n) {
                if (gc0 < n && (live0 ? r <= i : r < pposl[lc0])) A[r + (long)n * gc0] = a[0][0];
                if (gc0 + 1 < n && (live1 ? r <= i : r < pposl[lc0 + 1])) A[r + (long)n * (gc0 + 1)] = a[1][0];
            }
            if ((i & 15) == 15) {
#pragma unroll
                for (int j = 0; j < 15; ++j) { a[0][j] = a[0][j + 1]; a[1][j] = a[1][j + 1]; }
            }
        }
    }
}

int launch_qrcp_coop(Mat A, QrWork w, int n, int n_chains, hipStream_t s) {
    if (n > 256) { set_error("cooperative QRCP supports n <= 256"); return -1; }
    if (!w.sync || w.sync_stride < 2L * QC_P * QC_REC || !w.abort_words) { set_error("cooperative QRCP needs its sync workspace"); return -1; }
    DQ_HIP(hipMemsetAsync(w.sync, 0, sizeof(unsigned long long) * w.sync_stride * n_chains, s));
    DQ_HIP(hipMemsetAsync(w.abort_words, 0, sizeof(int) * n_chains, s));
    hipLaunchKernelGGL(qrcp_coop_kernel, dim3(QC_P, n_chains), dim3(QC_T), 0, s, A, w, n, (unsigned long long*)w.sync, w.sync_stride, w.abort_words, w.info);
    DQ_HIP(hipGetLastError());
    return 0;
}

long qrcp_coop_sync_granules() { return 2L * QC_P * QC_REC; }

}  // namespace dq
