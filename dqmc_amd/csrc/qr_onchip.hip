// qr_onchip.hip -- column-pivoted Householder QR for n <= 256 with the whole
// matrix resident in ONE compute unit's registers + LDS (no HBM/L2 traffic in
// the 256-step serial loop).  Same algorithm and output format as the
// streaming qrcp_kernel of qr.hip (LAPACK dlaqp2 semantics, reflectors and R0
// left in place in A, jpvt/tau out); to_LDR = source/stablelinalg.cpp:35-55.
//
// Why: the streaming kernel is bound by what one CU can pull through its L1
// (~64 B/clk): ~8*n^3 bytes per factorisation = 3.6 ms at n = 256, 40 % of a
// sweep.  512 KiB of fp64 fits one CU: 512 threads x 96 doubles in VGPRs
// (384 KiB) + 128 KiB of the 160 KiB LDS.
//
// Data layout (one 512-thread workgroup per chain, 8 waves, 2 per SIMD):
//   lane = 16*cl + rg, wave w:  thread owns columns c = 32w + 8cl + kc (kc=0..7)
//   and rows r = 16*b + rg (row block b = 0..15).  kc = 0..5 live in registers
//   a[kc][b], kc = 6,7 in LDS (lcol[kc-6][b][thread], conflict-free).
//   * the 16 row groups of a column sit in ONE DPP row of 16 lanes, so the
//     Householder dot products reduce with 4 DPP steps: no LDS, no barrier;
//   * a wave owns its 32 columns completely, so the update phase needs no
//     cross-wave communication at all;
//   * each v_r fetched from LDS feeds 8 columns (16 FMAs): LDS traffic stays far
//     under the fp64 FMA time;
//   * the step loop is unrolled over the 16 row blocks at compile time
//     (qr_block<JB>): inside block JB every register index is static, row i
//     (needed for the norm down-date) is register a[kc][JB], and the registers
//     of finished row blocks are dead, which hands them back to the allocator --
//     the register file is the scarce resource here (192 of 256 VGPRs hold data).
// Pivot choice uses a 64-bit key (norm bits with the low 8 mantissa bits
// replaced by 255 - column): columns whose norms agree to 2^-44 are ordered by
// index, exactly equal norms pick the lowest index as LAPACK does.
#include "common.h"
#include "wave.h"

namespace dq {

namespace {

constexpr int OC_N = 256;      // padded matrix size
constexpr int OC_T = 512;      // threads
constexpr int OC_KC = 8;       // columns per thread
constexpr int OC_KR = 6;       // ... of which in registers

__device__ __forceinline__ unsigned long long norm_key(double nrm, int c) {
    return (1ULL << 63) | ((unsigned long long)__double_as_longlong(nrm) & ~0xFFULL) | (unsigned long long)(255 - c);
}

struct OcShared {
    double* lcol;                 // [2][16][512]  columns kc = 6,7 of every thread
    double* xbuf;                 // [256] pivot column image
    double* vbuf;                 // [256] Householder vector
    double* vn1;                  // [256]
    double* vn2;                  // [256]
    unsigned long long* keys;     // [8]
    double* scal;                 // [0] = tau, [1] = beta, [2] = 1/(alpha-beta) of the current step
    double* rowi;                 // [256] row i of every column (norm down-date)
    int* pposl;                   // [256] pivot position of a pivoted column
};

#define LCOL(kc, blk) sh.lcol[(((kc) - OC_KR) * 16 + (blk)) * OC_T + t]
// Column map.  Thread pair index u = 4*wave + cl (0..31).  The two LDS-resident columns of a thread are the LOW
// column indices, c = 2u + (kc-6) in 0..63; the six register columns are c = 64 + 6u + kc.  In the matrices this
// code factors, (Bbar L) diag(d), column norms fall with the column index (d is sorted by the previous QRCP), so
// low columns are pivoted -- and go dead -- within the first ~64 steps: the LDS traffic (whose write bandwidth
// paced the update phase) disappears for the remaining three quarters of the factorisation.  Any matrix is still
// factored correctly; the map only decides which columns pay LDS latency.
#define COLOF(kc) ((kc) < OC_KR ? 64 + 6 * cbase + (kc) : 2 * cbase + ((kc) - OC_KR))      /* cbase = pair index u here */
#ifdef DQ_QR_STAMPS
#define QSTAMP(v) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); v = _t; }
#define QACC(slot, a, b) prof[slot] += (b) - (a);
#else
#define QSTAMP(v)
#define QACC(slot, a, b)
#endif

// All steps i in row block JB (i = 16*JB .. min(16*JB+15, n-1)).
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence = s_waitcnt vmcnt(0): with the
// reflector / R0 stores to HBM issued every step, each of the three barriers of a step waited for a store round trip.
// Those stores are consumed by other kernels only (formq, assemble_r), never by this workgroup through memory.
__device__ __forceinline__ void oc_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int JB>
__device__ __forceinline__ void qr_block(double (&a)[OC_KR][16], int& livem, unsigned& wlive, const OcShared& sh, double* __restrict__ A,
                                         double* tau, int* jpvt, int n, int t, int lane, int wave, int rg, int cl, int cbase
#ifdef DQ_QR_STAMPS
                                         , unsigned long long (&prof)[8]
#endif
                                         ) {
    constexpr int NJ = 16 - JB;                                     // live row blocks JB .. 15
    const double tol3z = 1.0536712127723509e-08;                    // sqrt(2^-53)
    const int i_end = min(16 * JB + 16, n);
    for (int i = 16 * JB; i < i_end; ++i) {
        // ---- [A] pivot + publish its column ----
#ifdef DQ_QR_STAMPS
        unsigned long long q0, q1, q2, q3, q4, q5, q6; QSTAMP(q0)
#endif
        unsigned long long best = sh.keys[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) { const unsigned long long o = sh.keys[k]; best = o > best ? o : best; }
        const int p = 255 - (int)(best & 0xFFULL);
        // owner of column p: pair index up, slot kcp; position qp of p among its wave's 32 columns
        const int up = p < 64 ? (p >> 1) : (p - 64) / 6;
        const int kcp = p < 64 ? OC_KR + (p & 1) : (p - 64) % 6;
        const int qp = p < 64 ? (p & 7) : 8 + (p - 64) % 24;
        if (wave == (up >> 2)) {               // wave-uniform: this wave holds the pivot column
            const bool mine = cl == (up & 3);
            const int r_i = i & 15;            // lane of a DPP row that holds row i (slot JB)
            double ss = 0.0, al_l = 0.0;
            // one direct copy per column slot (a wave-uniform switch): a select chain over the six register
            // columns costs ~12 v_cndmask per row block and sat on the critical path of every step
#define DQ_PUBLISH(EXPR_J, EXPR_0)                                                        \
            {                                                                             \
                _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                          \
                    const int r = 16 * (JB + j) + rg;                                     \
                    const double x = EXPR_J;                                              \
                    if (mine) sh.xbuf[r] = x;                                             \
                    if (r > i) ss += x * x;                                               \
                }                                                                         \
                al_l = EXPR_0;                                                            \
            }
            switch (kcp) {
                case 0: DQ_PUBLISH(a[0][JB + j], a[0][JB]) break;
                case 1: DQ_PUBLISH(a[1][JB + j], a[1][JB]) break;
                case 2: DQ_PUBLISH(a[2][JB + j], a[2][JB]) break;
                case 3: DQ_PUBLISH(a[3][JB + j], a[3][JB]) break;
                case 4: DQ_PUBLISH(a[4][JB + j], a[4][JB]) break;
                case 5: DQ_PUBLISH(a[5][JB + j], a[5][JB]) break;
                case 6: DQ_PUBLISH(LCOL(6, JB + j), LCOL(6, JB)) break;
                default: DQ_PUBLISH(LCOL(7, JB + j), LCOL(7, JB)) break;
            }
#undef DQ_PUBLISH
            ss = row16_sum(ss);                // every DPP row sums its own lanes; only the pivot's row is used
            // dlarfg scalars, computed once here (sqrt and two divisions cost ~0.4 us when all 8 waves repeat them)
            if (mine && rg == r_i) {
                double tau_l = 0.0, beta_l = al_l, scale_l = 0.0;
                if (ss != 0.0) {
                    beta_l = -copysign(sqrt(al_l * al_l + ss), al_l);
                    tau_l = (beta_l - al_l) / beta_l;
                    scale_l = 1.0 / (al_l - beta_l);
                }
                sh.scal[0] = tau_l; sh.scal[1] = beta_l; sh.scal[2] = scale_l;
            }
        }
        QSTAMP(q1)
        oc_barrier();
        QSTAMP(q2)
        // ---- [B] Householder vector (dlarfg) ----
        const double tau_i = sh.scal[0], beta = sh.scal[1], scale = sh.scal[2];
        if (t < OC_N && t >= 16 * JB) {
            const int r = t;
            double vr = 0.0;
            if (r == i) { vr = 1.0; if (r < n) A[r + (long)n * p] = beta; }
            else if (r > i) { vr = sh.xbuf[r] * scale; if (r < n) A[r + (long)n * p] = vr; }
            sh.vbuf[r] = vr;
        }
        if (t == 0) { tau[i] = tau_i; jpvt[i] = p; sh.pposl[p] = i; }
        if (up == cbase) livem &= ~(1 << kcp);
        if ((up >> 2) == wave) wlive &= ~(1u << qp);
        oc_barrier();
        QSTAMP(q3)
        // ---- [C] apply H to my live columns: two passes over the live row blocks, two blocks per trip ----
        {
            const double* vb = sh.vbuf + 16 * JB + rg;
            double s[OC_KC];
#pragma unroll
            for (int kc = 0; kc < OC_KC; ++kc) s[kc] = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; j += 2) {
                const double v0 = vb[16 * j];
                const double v1 = (j + 1 < NJ) ? vb[16 * (j + 1)] : 0.0;
                const double l60 = LCOL(6, JB + j), l70 = LCOL(7, JB + j);
                const double l61 = (j + 1 < NJ) ? LCOL(6, JB + j + 1) : 0.0, l71 = (j + 1 < NJ) ? LCOL(7, JB + j + 1) : 0.0;
#pragma unroll
                for (int kc = 0; kc < OC_KR; ++kc) { s[kc] += a[kc][JB + j] * v0; if (j + 1 < NJ) s[kc] += a[kc][JB + j + 1] * v1; }
                s[6] += l60 * v0 + l61 * v1; s[7] += l70 * v0 + l71 * v1;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kc = 0; kc < OC_KC; ++kc) { s[kc] = row16_sum(s[kc]) * tau_i; if (!(livem & (1 << kc))) s[kc] = 0.0; }
            QSTAMP(q4)
#pragma unroll
            for (int j = 0; j < NJ; j += 2) {
                const double v0 = vb[16 * j];
                const double v1 = (j + 1 < NJ) ? vb[16 * (j + 1)] : 0.0;
#pragma unroll
                for (int kc = 0; kc < OC_KR; ++kc) { a[kc][JB + j] -= s[kc] * v0; if (j + 1 < NJ) a[kc][JB + j + 1] -= s[kc] * v1; }
                if (livem & 64) { LCOL(6, JB + j) -= s[6] * v0; if (j + 1 < NJ) LCOL(6, JB + j + 1) -= s[6] * v1; }
                if (livem & 128) { LCOL(7, JB + j) -= s[7] * v0; if (j + 1 < NJ) LCOL(7, JB + j + 1) -= s[7] * v1; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        QSTAMP(q5)
        // ---- norm down-date (dlaqp2), new keys ----
        // The lanes that hold row i (rg == i & 15, register a[kc][JB]) drop their 8 row-i entries in LDS;
        // then lane q < 32 of every wave owns the wave's q-th column (q < 8: LDS columns 8w+q, else 64+24w+q-8) and runs ONE divide/sqrt chain
        // (eight sequential chains in four lanes cost > 1 us per step).  A column whose norm must be
        // recomputed (cancellation) is flagged and handled cooperatively by its 16 row-group lanes (rare).
        if (rg == (i & 15)) {
#pragma unroll
            for (int kc = 0; kc < OC_KC; ++kc) sh.rowi[COLOF(kc)] = kc < OC_KR ? a[kc < OC_KR ? kc : 0][JB] : LCOL(kc, JB);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): my wave's LDS writes are done (wave-local hand-off)
        __builtin_amdgcn_wave_barrier();
        int need = 0;                                          // bit kc (of MY columns) -> recompute
        double nn = 0.0; bool col_live = false; int need_col = 0;
        const int ql = lane & 31;                              // lane q < 32 owns the q-th column of this wave
        const int mycol = ql < 8 ? 8 * wave + ql : 64 + 24 * wave + (ql - 8);
        col_live = lane < 32 && ((wlive >> (lane & 31)) & 1u);      // wlive: wave-uniform mask of this wave's live columns
        if (lane < 32 && col_live) {
            const double rv = sh.rowi[mycol];
            const double n1_ = sh.vn1[mycol];
            nn = n1_;
            if (n1_ != 0.0) {
                double temp = fabs(rv) / n1_; temp = fmax(0.0, 1.0 - temp * temp);
                const double rr = n1_ / sh.vn2[mycol];
                if (temp * rr * rr <= tol3z) need_col = 1;
                else { nn = n1_ * sqrt(temp); sh.vn1[mycol] = nn; }
            }
        }
        const unsigned long long needmask = __ballot(need_col);   // bit q: column 32*wave + q needs a recompute
        if (needmask != 0ULL) {
            // my thread's columns sit at wave positions q = 8 + 6*cl + kc (register slots) and q = 2*cl + kc - 6 (LDS slots)
            need = (int)((needmask >> (8 + 6 * cl)) & 0x3FULL) | (int)(((needmask >> (2 * cl)) & 0x3ULL) << 6);
#pragma unroll
            for (int kc = 0; kc < OC_KC; ++kc) {
                const unsigned long long rows_kc = kc < OC_KR ? (0x04104100ULL << kc)                     /* q = 8+kc, 14+kc, 20+kc, 26+kc */
                                                             : (0x55ULL << (kc - OC_KR));                  /* q = kc-6, +2, +4, +6 */
                if (needmask & rows_kc) {                          // some DPP row of this wave needs column slot kc
                    double tl = 0.0;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (16 * (JB + j) + rg > i) {
                            const double x = kc < OC_KR ? a[kc < OC_KR ? kc : 0][JB + j] : LCOL(kc, JB + j);
                            tl += x * x;
                        }
                    }
                    tl = row16_sum(tl);
                    if (rg == 0 && ((need >> kc) & 1)) {
                        const double v2 = (i + 1 < n) ? sqrt(tl) : 0.0;
                        sh.vn1[COLOF(kc)] = v2; sh.vn2[COLOF(kc)] = v2;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (lane < 32 && need_col) nn = sh.vn1[mycol];
        }
        {
            unsigned long long key = (lane < 32 && col_live) ? norm_key(nn, mycol) : 0ULL;
            key = wave_max_u64(key);
            if (lane == 0) sh.keys[wave] = key;
        }
        QSTAMP(q6)
        QACC(0, q0, q1) QACC(1, q1, q2) QACC(2, q2, q3) QACC(3, q3, q4) QACC(4, q4, q5) QACC(5, q5, q6)
        // ---- end of the row block: it is final -> store it as R0 ----
        if (i == i_end - 1) {
            const int r = 16 * JB + rg;
            if (r < n) {
                // live columns: every row <= i is an R0 entry.  Pivoted columns: rows above the pivot
                // position are R0 entries; the diagonal (beta) and the reflector below it are already in A.
#pragma unroll
                for (int kc = 0; kc < OC_KC; ++kc) {
                    if (COLOF(kc) < n && ((livem & (1 << kc)) ? r <= i : r < sh.pposl[COLOF(kc)]))
                        A[r + (long)n * COLOF(kc)] = kc < OC_KR ? a[kc < OC_KR ? kc : 0][JB] : LCOL(kc, JB);
                }
            }
        }
        oc_barrier();
    }
}

}  // namespace

__global__ __launch_bounds__(OC_T) void qrcp_onchip_kernel(Mat Am, QrWork w, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    OcShared sh;
    sh.lcol = reinterpret_cast<double*>(smem);
    sh.xbuf = sh.lcol + 2 * 16 * OC_T;
    sh.vbuf = sh.xbuf + OC_N;
    sh.vn1 = sh.vbuf + OC_N;
    sh.vn2 = sh.vn1 + OC_N;
    sh.keys = reinterpret_cast<unsigned long long*>(sh.vn2 + OC_N);
    sh.scal = reinterpret_cast<double*>(sh.keys + 8);
    sh.rowi = sh.scal + 8;
    sh.pposl = reinterpret_cast<int*>(sh.rowi + OC_N);

    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    double* tau = w.tau + (long)chain * w.tau_stride;
    int* jpvt = w.jpvt + (long)chain * w.jpvt_stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int rg = lane & 15, cl = lane >> 4;
    const int cbase = 4 * wave + cl;                                // pair index u of the column map (COLOF)

    double a[OC_KR][16];
    int livem = 0;                                                  // bit kc: my column kc has not been pivoted yet
#pragma unroll
    for (int kc = 0; kc < OC_KC; ++kc) if (COLOF(kc) < n) livem |= 1 << kc;        // padding columns are never live
    if (t < OC_N) sh.pposl[t] = OC_N;
    unsigned wlive = 0;                                             // bit q: column 32*wave + q is live (wave-uniform)
    for (int q = 0; q < 32; ++q) { const int c = q < 8 ? 8 * wave + q : 64 + 24 * wave + (q - 8); if (c < n) wlive |= 1u << q; }

    // ---- load, initial norms, initial keys ----
    {
        double nrm[OC_KC];
#pragma unroll
        for (int kc = 0; kc < OC_KC; ++kc) nrm[kc] = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int r = 16 * j + rg;
#pragma unroll
            for (int kc = 0; kc < OC_KC; ++kc) {
                const double x = (r < n && COLOF(kc) < n) ? A[r + (long)n * COLOF(kc)] : 0.0;
                if (kc < OC_KR) a[kc < OC_KR ? kc : 0][j] = x; else LCOL(kc, j) = x;
                nrm[kc] += x * x;
            }
        }
        unsigned long long key = 0;
#pragma unroll
        for (int kc = 0; kc < OC_KC; ++kc) {
            const double nn = sqrt(row16_sum(nrm[kc]));
            if (rg == 0) {
                sh.vn1[COLOF(kc)] = nn; sh.vn2[COLOF(kc)] = nn;
                if (livem & (1 << kc)) { const unsigned long long k = norm_key(nn, COLOF(kc)); key = k > key ? k : key; }
            }
        }
        key = wave_max_u64(key);
        if (lane == 0) sh.keys[wave] = key;
    }
    __syncthreads();

#ifdef DQ_QR_STAMPS
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tk0; QSTAMP(tk0)
#define DQ_QR_BLOCK(JB) if (16 * JB < n) qr_block<JB>(a, livem, wlive, sh, A, tau, jpvt, n, t, lane, wave, rg, cl, cbase, prof);
#else
#define DQ_QR_BLOCK(JB) if (16 * JB < n) qr_block<JB>(a, livem, wlive, sh, A, tau, jpvt, n, t, lane, wave, rg, cl, cbase);
#endif
    DQ_QR_BLOCK(0) DQ_QR_BLOCK(1) DQ_QR_BLOCK(2) DQ_QR_BLOCK(3) DQ_QR_BLOCK(4) DQ_QR_BLOCK(5) DQ_QR_BLOCK(6) DQ_QR_BLOCK(7)
    DQ_QR_BLOCK(8) DQ_QR_BLOCK(9) DQ_QR_BLOCK(10) DQ_QR_BLOCK(11) DQ_QR_BLOCK(12) DQ_QR_BLOCK(13) DQ_QR_BLOCK(14) DQ_QR_BLOCK(15)
#undef DQ_QR_BLOCK
#ifdef DQ_QR_STAMPS
    if ((t & 63) == 0 && (wave == 0 || wave == 7)) {
        unsigned long long tk1; QSTAMP(tk1)
        printf("qr wave %d n=%d total %llu | A(pivot+publish) %llu | barrier1 %llu | B %llu | C1(dots) %llu | C2(axpy) %llu | norms+keys %llu\n",
               wave, n, tk1 - tk0, prof[0], prof[1], prof[2], prof[3], prof[4], prof[5]);
    }
#endif
}
#undef LCOL
#undef COLOF

int qr_onchip_init_device() {
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qrcp_onchip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

int launch_qrcp_onchip(Mat A, QrWork w, int n, int n_chains, hipStream_t s) {
    if (n > OC_N) { set_error("on-chip QRCP supports n <= 256"); return -1; }
    const size_t lds = sizeof(double) * (2 * 16 * OC_T + 5 * OC_N + 8 + 8) + sizeof(int) * OC_N + 64;
    hipLaunchKernelGGL(qrcp_onchip_kernel, dim3(1, n_chains), dim3(OC_T), lds, s, A, w, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
