// qr_colown.hip -- column-pivoted Householder QR for n <= 256, "column-owner" layout, one workgroup per chain.
// Same algorithm and output format as the streaming qrcp_kernel of qr.hip and the cooperative kernel of qr_coop.hip (LAPACK
// dlaqp2 semantics: reflectors and R0 left in place in A without column swaps, jpvt / tau out); to_LDR = source/stablelinalg.cpp:35-55.
//
// Why this layout.  A 2-D block-cyclic single-workgroup kernel (round 1's qr_onchip.hip, deleted) spent ~11500 clk per step (stamps): every
// phase pays for cross-lane traffic -- the pivot column is gathered through LDS, each v_r is an LDS broadcast read
// (a broadcast still occupies the LDS pipe for the full 64 lanes), dot products end in DPP row reductions, a quarter
// of the matrix is read twice and written once in LDS per step, and ~285 VGPRs spill.  Here
//   * a thread owns (a row range of) ONE column: dot products v^T a_c and the norm down-dates are thread-local,
//     no reduction at all; two threads share a column (rows split), so the only exchange is one partial dot per
//     column per step through LDS;
//   * the Householder vector is broadcast by the DPP network: every 16-lane row of every wave holds x[16c + r] in
//     lane r of register c, and the FMAs take it with row_newbcast:r (v_fmac_f64_dpp, CDNA's 64-bit DPP mode) --
//     one instruction per (row, column) element for the dot and one for the update, nothing else;
//   * storage is tiered by row LIFETIME (row k is final after step k): rows 0..63 live in LDS (128 KiB, dead after
//     the first quarter of the steps), rows 64..255 in registers as six d16 vectors per thread (192 of 256 VGPRs);
//     a dead 16-row block costs nothing any more because the step loop is expanded per row block (CO_BLOCK(JB)) and
//     blocks below JB are not even compiled in;
//   * the dots run on the UNSCALED pivot column x (v = x * scale, v_k = 1): v^T a = a_k + scale * (x^T a), so the
//     dlarfg scalars (one sqrt, two divisions) are computed by a single wave while all waves do the dots;
//   * barriers order LDS only (s_waitcnt lgkmcnt(0) + s_barrier): the reflector / tau / jpvt stores to HBM are
//     fire-and-forget.
// Thread map (512 threads = 8 waves, two per SIMD): col = t & 255, half = t >> 8.
//   half h: the 16-row blocks h, h + 2 in LDS and 4 + h, 6 + h, ..., 14 + h in registers (interleaved: see CO_ACT)
// The partial column norms vn1 / vn2 live in LDS and are down-dated by half 0 at every step (the half with less FMA work: its row
// blocks die one block earlier); the half that owns row k ("active" half of step k) only stores R0(k, :).
// Register file by hand.  192 of the 256 VGPRs of a thread hold matrix rows for the whole kernel, and the register
// allocator cannot be talked into that: as C arrays or vector types the rows were spilled and reloaded around every
// asm block (2755 .. 10139 spills, 2.7 ms per factorisation).  So the matrix registers are taken out of its hands:
// the kernel is compiled with amdgpu_num_vgpr(32), which confines the allocator to v0..v63, and rows live in
// v64..v255 (block c, row r in v[64 + 32c + 2r : +1]), touched only by the generated inline-asm accessors of
// qr_colown_regs.inc (scripts/gen_qr_colown_regs.py): DPP dot / axpy blocks, LDS publish, element moves.
#include "common.h"
#include "wave.h"
#include "qr_colown_regs.inc"
// the accessors name v64..v255 in their clobber lists on purpose: those registers are reserved FROM the allocator FOR them
#pragma clang diagnostic ignored "-Winline-asm"

namespace dq {

namespace {

constexpr int CO_N = 256;      // padded matrix size
constexpr int CO_T = 512;      // threads

__device__ __forceinline__ unsigned long long co_norm_key(double nrm, int c) {
    return (1ULL << 63) | ((unsigned long long)__double_as_longlong(nrm) & ~0xFFULL) | (unsigned long long)(255 - c);
}
// lane id, recomputed wherever it is needed: as an ordinary value the compiler keeps threadIdx-derived indices alive
// across the whole kernel, and with only 64 VGPRs to work with it spills exactly those and reloads them from scratch
// (s_waitcnt vmcnt(0)) on the critical path of every step
__device__ __forceinline__ int co_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ void co_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// 8-row chunk of an LDS-resident block through 8 temporaries: d_{r & 3} += x[lane R] * T[r]
#define CO_DOT8(X, V, R0, R1, R2, R3, R4, R5, R6, R7)                                                              \
    asm volatile("s_nop 4\n\t"                                                                                     \
        "v_fmac_f64_dpp %0, %4, %5 row_newbcast:" #R0 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %1, %4, %6 row_newbcast:" #R1 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %2, %4, %7 row_newbcast:" #R2 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %3, %4, %8 row_newbcast:" #R3 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %0, %4, %9 row_newbcast:" #R4 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %1, %4, %10 row_newbcast:" #R5 " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_fmac_f64_dpp %2, %4, %11 row_newbcast:" #R6 " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_fmac_f64_dpp %3, %4, %12 row_newbcast:" #R7 " row_mask:0xf bank_mask:0xf"                               \
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)                                                                   \
        : "v"(X), "v"(V[0]), "v"(V[1]), "v"(V[2]), "v"(V[3]), "v"(V[4]), "v"(V[5]), "v"(V[6]), "v"(V[7]))
// T[r] += x[lane R] * MW
#define CO_AXPY8(V, X, MW, R0, R1, R2, R3, R4, R5, R6, R7)                                                         \
    asm volatile("s_nop 4\n\t"                                                                                     \
        "v_fmac_f64_dpp %0, %8, %9 row_newbcast:" #R0 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %1, %8, %9 row_newbcast:" #R1 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %2, %8, %9 row_newbcast:" #R2 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %3, %8, %9 row_newbcast:" #R3 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %4, %8, %9 row_newbcast:" #R4 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %5, %8, %9 row_newbcast:" #R5 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %6, %8, %9 row_newbcast:" #R6 " row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %7, %8, %9 row_newbcast:" #R7 " row_mask:0xf bank_mask:0xf"                                \
        : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7])            \
        : "v"(X), "v"(MW))

#ifdef DQ_QR_STAMPS
#define CST(v) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); v = _t; }
#define CACC(i, a, b) prof[i] += b - a;
#define CST_DECL unsigned long long c0, c1, c2, c3, c4, c5, c6, c7;
#define CST_TAIL { unsigned long long c8; CST(c8) CACC(7, c7, c8) }
#else
#define CST(v)
#define CACC(i, a, b)
#define CST_DECL
#define CST_TAIL
#endif

struct CoShared {
    double* alds;                 // [64][256]  rows 0..63 of the matrix, alds[row * 256 + col]
    double* vbuf;                 // [256] the pivot column x (rows >= k)
    double* pdot;                 // [2][256] partial x^T a_c of the two halves (also: partial sums of squares)
    double* rowk;                 // [256] row k of the matrix before the reflector is applied
    double* vn;                   // [2][256] partial column norms vn1 / vn2 (dlaqp2), used by the half that owns row k
    double* taul;                 // [256] tau, written to memory at the end
    int* jpl;                     // [256] jpvt
    double* vscale;               // [256] per column: the factor that turns the stored x (rows below the pivot position) into the reflector v
    int* posc;                    // [256] per column: its pivot position
    unsigned long long* keys;     // [4] pivot candidates of the four waves of the half that owns the current row (+4 unused)
    double* scal;                 // [4] tau, beta, scale
    unsigned int* needany;        // [1] some column needs its norm recomputed
    unsigned char* needc;         // [256] which
    unsigned char* livec;         // [256] column not pivoted yet
};

}  // namespace

// which half owns row k of row block JB: even blocks -> half 0, odd blocks -> half 1, in LDS (blocks 0..3) and in registers (4..15)
// alike.  Row k is final after step k, so a half loses a 16-row block every 32 steps: with the blocks INTERLEAVED both halves
// shrink at the same rate and the dots / update phases of the two waves of a SIMD stay balanced (contiguous halves left the
// first half idle ~1 000 clk per step waiting for the second at the end-of-step barrier).
#define CO_ACT(JB) ((JB) & 1)

__global__ __launch_bounds__(CO_T) __attribute__((amdgpu_num_vgpr(32))) void qrcp_colown_kernel(Mat Am, QrWork w, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    CoShared sh;
    sh.alds = reinterpret_cast<double*>(smem);
    sh.vbuf = sh.alds + 64 * CO_N;
    sh.pdot = sh.vbuf + CO_N;
    sh.rowk = sh.pdot + 2 * CO_N;
    sh.vn = sh.rowk + CO_N;
    sh.taul = sh.vn + 2 * CO_N;
    sh.jpl = reinterpret_cast<int*>(sh.taul + CO_N);
    sh.posc = sh.jpl + CO_N;
    sh.vscale = reinterpret_cast<double*>(sh.posc + CO_N);
    sh.keys = reinterpret_cast<unsigned long long*>(sh.vscale + CO_N);
    sh.scal = reinterpret_cast<double*>(sh.keys + 8);
    sh.needany = reinterpret_cast<unsigned int*>(sh.scal + 4);
    sh.needc = reinterpret_cast<unsigned char*>(sh.needany + 4);
    sh.livec = sh.needc + CO_N;
    // LDS byte address of vbuf for the ds_write_b128 of the publish blocks
    const unsigned vbuf_lds = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)smem) + 64u * CO_N * 8u;

    const int chain = blockIdx.y;
    double* __restrict__ A = Am.at(chain);
    double* tau = w.tau + (long)chain * w.tau_stride;
    int* jpvt = w.jpvt + (long)chain * w.jpvt_stride;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), half = wave >> 2;     // wave-uniform: everything derived from them lives in SGPRs
    // first column of this wave.  The two owners of a column (one per half) sit on DIFFERENT SIMDs (wave w runs on SIMD w & 3): the
    // publish of the pivot column is 48 single-lane ds_write_b128 per owner, and two waves of one SIMD issue them one after the other
    const int cw = (((wave & 3) + half) & 3) * 64;
#define CO_IDS const int lane = co_lane(), col = cw + lane, r16 = lane & 15, t = wave * 64 + lane; (void)t; (void)r16;
    const double tol3z = 1.0536712127723509e-08; // sqrt(2^-53)


    // ---- load: registers, LDS rows, initial norms ----
    {
        CO_IDS
        const bool live = col < n;                   // padding columns are never live
        if (half == 0) sh.livec[col] = live ? 1 : 0;
        double ssq = 0.0;
#define CO_LOADV(C) ({ const int row = 16 * (4 + 2 * (C) + half) + r; const double x = (row < n && col < n) ? A[row + (long)n * col] : 0.0; ssq += x * x; x; })
        CO_RFOREACH_WR_0(CO_LOADV(0)) CO_RFOREACH_WR_1(CO_LOADV(1)) CO_RFOREACH_WR_2(CO_LOADV(2))
        CO_RFOREACH_WR_3(CO_LOADV(3)) CO_RFOREACH_WR_4(CO_LOADV(4)) CO_RFOREACH_WR_5(CO_LOADV(5))
#undef CO_LOADV
#pragma unroll 4
        for (int r = 0; r < 32; ++r) {
            const int row = 16 * (half + 2 * (r >> 4)) + (r & 15);            // LDS row blocks half, half + 2
            const double x = (row < n && col < n) ? A[row + (long)n * col] : 0.0;
            sh.alds[row * CO_N + col] = x; ssq += x * x;
        }
        sh.pdot[half * CO_N + col] = ssq;
        if (t == 0) sh.needany[0] = 0u;
        __syncthreads();
        const double nn = sqrt(sh.pdot[col] + sh.pdot[CO_N + col]);
        if (half == 0) { sh.vn[col] = nn; sh.vn[CO_N + col] = nn; }
        unsigned long long key = (half == 0 && live) ? co_norm_key(nn, col) : 0ULL;
        key = wave_max_u64(key);
        if (lane == 0 && half == 0) sh.keys[wave] = key;
        __syncthreads();
    }

    // x registers of one 16-row block GB for step k: lane r of every 16-lane row holds x[16 GB + r], rows <= k zeroed
#define CO_X(GB) ((16 * (GB) + r16 > k) ? sh.vbuf[16 * (GB) + r16] : 0.0)
#define CO_LDS_DOT(GB)                                                                                             \
    {                                                                                                              \
        const double x = CO_X(GB);                                                                                 \
        { double T[8]; _Pragma("unroll") for (int r = 0; r < 8; ++r) T[r] = sh.alds[(16 * (GB) + r) * CO_N + col];     \
          CO_DOT8(x, T, 0, 1, 2, 3, 4, 5, 6, 7); }                                                                 \
        { double T[8]; _Pragma("unroll") for (int r = 0; r < 8; ++r) T[r] = sh.alds[(16 * (GB) + 8 + r) * CO_N + col]; \
          CO_DOT8(x, T, 8, 9, 10, 11, 12, 13, 14, 15); }                                                           \
    }
#define CO_LDS_AXPY(GB)                                                                                            \
    {                                                                                                              \
        const double x = CO_X(GB);                                                                                 \
        { double T[8]; _Pragma("unroll") for (int r = 0; r < 8; ++r) T[r] = sh.alds[(16 * (GB) + r) * CO_N + col];     \
          CO_AXPY8(T, x, mw, 0, 1, 2, 3, 4, 5, 6, 7);                                                              \
          _Pragma("unroll") for (int r = 0; r < 8; ++r) sh.alds[(16 * (GB) + r) * CO_N + col] = T[r]; }               \
        { double T[8]; _Pragma("unroll") for (int r = 0; r < 8; ++r) T[r] = sh.alds[(16 * (GB) + 8 + r) * CO_N + col]; \
          CO_AXPY8(T, x, mw, 8, 9, 10, 11, 12, 13, 14, 15);                                                        \
          _Pragma("unroll") for (int r = 0; r < 8; ++r) sh.alds[(16 * (GB) + 8 + r) * CO_N + col] = T[r]; }           \
    }
#define CO_LDS_SCALE(GB) { _Pragma("unroll") for (int r = 0; r < 16; ++r) { if (16 * (GB) + r > k) sh.alds[(16 * (GB) + r) * CO_N + col] *= scale; } }
#define CO_LDS_SUMSQ(GB) { _Pragma("unroll") for (int r = 0; r < 16; ++r) { const double x = (16 * (GB) + r > k) ? sh.alds[(16 * (GB) + r) * CO_N + col] : 0.0; psq += x * x; } }
    // register block C of a half is global row block 4 + 2 C (half 0, tag GB0 = 4) or 5 + 2 C (half 1, tag GB0 = 10)
#define CO_GB(GB0, C) ((GB0) == 4 ? 4 + 2 * (C) : 5 + 2 * (C))
#define CO_REG_X(C, GB0) const double x##C = (CO_GB(GB0, C) >= (JB_)) ? CO_X(CO_GB(GB0, C)) : 0.0;
#define CO_REG_DOT(C, GB0) if (CO_GB(GB0, C) >= (JB_)) { CO_RDOT_##C(x##C) }
#define CO_REG_AXPY(C, GB0) if (CO_GB(GB0, C) >= (JB_)) { CO_RAXPY_##C(x##C, mw) }
#define CO_REG_SCALE(C, GB0) if (CO_GB(GB0, C) > (JB_)) CO_RFOREACH_MUL_##C(scale) else if (CO_GB(GB0, C) == (JB_)) CO_RFOREACH_MUL_##C((16 * (CO_GB(GB0, C)) + r > k) ? scale : 1.0)
#define CO_REG_PUB(C, GB0) if (CO_GB(GB0, C) >= (JB_)) { const unsigned a_ = vbuf_lds + 128u * (CO_GB(GB0, C)); CO_RPUB_##C(a_) }
#define CO_REG_SUMSQ(C, GB0) if (CO_GB(GB0, C) >= (JB_)) CO_RFOREACH_RD_##C({ const double x = (16 * (CO_GB(GB0, C)) + r > k) ? v : 0.0; psq += x * x; })
#define CO_REG_ALL(OP, GB0) OP(0, GB0) OP(1, GB0) OP(2, GB0) OP(3, GB0) OP(4, GB0) OP(5, GB0)

    // ---- the steps of row block JB (k = 16 JB .. 16 JB + 15): blocks below JB are dead and not compiled in ----
#define CO_BLOCK(JB)                                                                                                                  \
    if (16 * (JB) < n) {                                                                                                              \
        constexpr int JB_ = (JB);                                                                                                     \
        constexpr int act = CO_ACT(JB);                                                                                               \
        const int k_end = min(16 * (JB) + 16, n);                                                                                     \
        for (int k = 16 * (JB); k < k_end; ++k) {                                                                                     \
            CST_DECL CST(c0)                                                                \
            /* ---- pivot: largest partial column norm (the keys of the four waves of the half that owned row k - 1) ---- */          \
            CO_IDS                                                                                                                    \
            const bool upd = sh.livec[col] != 0;           /* the reflector touches live columns, the pivot column included */       \
            const unsigned long long ka = sh.keys[0], kb = sh.keys[1], kc = sh.keys[2], kd = sh.keys[3];              \
            unsigned long long kmax = ka > kb ? ka : kb; { const unsigned long long k2 = kc > kd ? kc : kd; kmax = k2 > kmax ? k2 : kmax; }   \
            const int p = 255 - (int)(kmax & 0xFFULL);                                                                                \
            const bool live = upd && col != p;                                                                                        \
            if (col == p) {                                /* its two owners publish their live rows */                               \
                                                                                                           \
                if (half == 0) { CO_REG_ALL(CO_REG_PUB, 4) } else { CO_REG_ALL(CO_REG_PUB, 10) }                                       \
            }                                                                                                                         \
            if ((JB) < 4 && wave == 1) { const int row = t - 64; if (row >= k) sh.vbuf[row] = sh.alds[row * CO_N + p]; }             \
            CST(c1)                                                                                                                   \
            co_barrier();                                                                                                             \
            CST(c2)                                                                                                                   \
            /* ---- partial dots x^T a_c over my live rows (rows > k), row k of the active half ---- */                                \
            double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;                                                                            \
            if (half == 0) {                                                                                                          \
                CO_REG_ALL(CO_REG_X, 4)                    /* all x loads of the phase in flight before the first FMA block */        \
                CO_REG_ALL(CO_REG_DOT, 4)                                                                                             \
                if (0 >= (JB)) CO_LDS_DOT(0) if (2 >= (JB)) CO_LDS_DOT(2)                                                             \
            } else {                                                                                                                  \
                CO_REG_ALL(CO_REG_X, 10)                                                                                              \
                CO_REG_ALL(CO_REG_DOT, 10)                                                                                            \
                if (1 >= (JB)) CO_LDS_DOT(1) if (3 >= (JB)) CO_LDS_DOT(3)                                                             \
            }                                                                                                                         \
            sh.pdot[half * CO_N + col] = (d0 + d1) + (d2 + d3);                                                                       \
            if (half == act) {                                                                                                        \
                double ak;                                                                                                            \
                if ((JB) < 4) ak = sh.alds[k * CO_N + col];                                                                           \
                else {                                                                                                                \
                    const int q = __builtin_amdgcn_readfirstlane(k & 15);                                                             \
                    if ((((JB) - 4) >> 1) == 0) { CO_RGET_0(q, ak) } else if ((((JB) - 4) >> 1) == 1) { CO_RGET_1(q, ak) }                  \
                    else if ((((JB) - 4) >> 1) == 2) { CO_RGET_2(q, ak) } else if ((((JB) - 4) >> 1) == 3) { CO_RGET_3(q, ak) }             \
                    else if ((((JB) - 4) >> 1) == 4) { CO_RGET_4(q, ak) } else { CO_RGET_5(q, ak) }                                      \
                }                                                                                                                     \
                sh.rowk[col] = ak;                                                                                                    \
            }                                                                                                                         \
            if (wave == 0) {                               /* dlarfg scalars, once (source of the formulas: LAPACK dlarfg) */         \
                double ss = 0.0;                                                                                                      \
                _Pragma("unroll") for (int gb = (JB); gb < 16; ++gb) { const double x = CO_X(gb); ss += x * x; }                      \
                ss = row16_sum(ss);                                                                                                   \
                const double alpha = sh.vbuf[k];                                                                                      \
                double tau_l = 0.0, beta_l = alpha, scale_l = 0.0;                                                                    \
                if (ss != 0.0) {                                                                                                      \
                    beta_l = -copysign(sqrt(alpha * alpha + ss), alpha);                                                              \
                    tau_l = (beta_l - alpha) / beta_l;                                                                                \
                    scale_l = 1.0 / (alpha - beta_l);                                                                                 \
                }                                                                                                                     \
                if (lane == 0) { sh.scal[0] = tau_l; sh.scal[1] = beta_l; sh.scal[2] = scale_l; }                                     \
            }                                                                                                                         \
            CST(c3)                                                                                                                   \
            co_barrier();                                                                                                             \
            CST(c4)                                                                                                                   \
            /* ---- apply H = I - tau v v^T, v = x * scale (v_k = 1): a_c -= v * w_c, w_c = tau (a_kc + scale x^T a_c) ---- */       \
            const double tau_k = sh.scal[0], beta = sh.scal[1], scale = sh.scal[2];                                                   \
            const double ak = sh.rowk[col];                                                                                           \
            const double wc = upd ? tau_k * (ak + scale * (sh.pdot[col] + sh.pdot[CO_N + col])) : 0.0;                                \
            const double mw = (col == p) ? 0.0 : -scale * wc;       /* the pivot column keeps x: it becomes the reflector below */    \
            CST(c5)                                                                                                                   \
            /* ---- row k is final: R0(k, c); partial column norms (dlaqp2 down-date), next pivot candidates.  BEFORE the update of the   \
               rows below: the latency-bound norm arithmetic of this half overlaps with the FMA blocks of the other half's wave on the same SIMD ---- */                  \
            bool need = false;                                                                                                        \
            const double rk = (col == p) ? beta : ak - wc;                                                                            \
            if (half == act) {                             /* the owner of row k stores R0(k, c) */                                   \
                if (upd) {                                                                                                            \
                    if ((JB) < 4) sh.alds[k * CO_N + col] = rk;                                                                       \
                    else {                                                                                                            \
                        const int q = __builtin_amdgcn_readfirstlane(k & 15);                                                         \
                        if ((((JB) - 4) >> 1) == 0) { CO_RSET_0(q, rk) } else if ((((JB) - 4) >> 1) == 1) { CO_RSET_1(q, rk) }              \
                        else if ((((JB) - 4) >> 1) == 2) { CO_RSET_2(q, rk) } else if ((((JB) - 4) >> 1) == 3) { CO_RSET_3(q, rk) }         \
                        else if ((((JB) - 4) >> 1) == 4) { CO_RSET_4(q, rk) } else { CO_RSET_5(q, rk) }                                  \
                    }                                                                                                                 \
                }                                                                                                                     \
            }                                                                                                                         \
            if (half == 0) {                               /* norm down-dates and pivot candidates: ALWAYS half 0 (every thread has a_kc and w_c); half 1 is the slower half in both FMA phases (stamps: dots 1 300 against 960 clk per step, update 1 850 against 1 550).  Measured: on the active half (alternating) 1 022 us per to_ldr, on half 0 957, split two waves + two waves 997 */ \
                double vn1 = sh.vn[col];                                                                                              \
                if (live && vn1 != 0.0) {                                                                                             \
                    double temp = fabs(rk) / vn1; temp = fmax(0.0, 1.0 - temp * temp);                                                \
                    const double rr = vn1 / sh.vn[CO_N + col];                                                                        \
                    if (temp * rr * rr <= tol3z) need = true;                                                                         \
                    else { vn1 = vn1 * sqrt(temp); sh.vn[col] = vn1; }                                                                \
                }                                                                                                                     \
                sh.needc[col] = need ? 1 : 0;                                                                                         \
                if (__ballot(need) != 0ULL && lane == 0) atomicOr(sh.needany, 1u);                                                    \
                unsigned long long key = live ? co_norm_key(vn1, col) : 0ULL;                                                         \
                key = wave_max_u64(key);                                                                                              \
                if (lane == 0) sh.keys[wave & 3] = key;                                                                               \
            }                                                                                                                         \
            if (half == 0) {                                                                                                          \
                CO_REG_ALL(CO_REG_X, 4)                                                                                               \
                CO_REG_ALL(CO_REG_AXPY, 4)                                                                                            \
                if (0 >= (JB)) CO_LDS_AXPY(0) if (2 >= (JB)) CO_LDS_AXPY(2)                                                           \
                                                                      \
            } else {                                                                                                                  \
                CO_REG_ALL(CO_REG_X, 10)                                                                                              \
                CO_REG_ALL(CO_REG_AXPY, 10)                                                                                           \
                if (1 >= (JB)) CO_LDS_AXPY(1) if (3 >= (JB)) CO_LDS_AXPY(3)                                                           \
                                                                     \
            }                                                                                                                         \
            /* the pivot column keeps x below row k; v = x * scale is formed when the matrix is written out (the scaling pass of   \
               the two owner waves was ~700 clk on the critical path of every step) */                                                  \
            if (t == 0) { sh.taul[k] = tau_k; sh.jpl[k] = p; sh.vscale[p] = scale; sh.posc[p] = k; }                                                                       \
            if (col == p && half == 0) sh.livec[col] = 0;  /* read again only after the barrier that ends the step */                                                                                                                         \
            CST(c6)                                                                                                                   \
            co_barrier();                                                                                                             \
            CST(c7)                                                                                                                   \
            CACC(0, c0, c1) CACC(1, c1, c2) CACC(2, c2, c3) CACC(3, c3, c4) CACC(4, c4, c5) CACC(5, c5, c6) CACC(6, c6, c7)           \
            if (sh.needany[0] != 0u) {                     /* dlaqp2's recomputation branch: exact norms of rows > k, both halves */ \
                const bool mine = sh.needc[col] != 0;                                                                                 \
                double psq = 0.0;                                                                                                     \
                if (mine) {                                                                                                           \
                    if (half == 0) {                                                                                                  \
                        if (0 >= (JB)) CO_LDS_SUMSQ(0) if (2 >= (JB)) CO_LDS_SUMSQ(2)                                                 \
                        CO_REG_ALL(CO_REG_SUMSQ, 4)                                                                                   \
                    } else {                                                                                                          \
                        if (1 >= (JB)) CO_LDS_SUMSQ(1) if (3 >= (JB)) CO_LDS_SUMSQ(3)                                                 \
                        CO_REG_ALL(CO_REG_SUMSQ, 10)                                                                                  \
                    }                                                                                                                 \
                }                                                                                                                     \
                sh.pdot[half * CO_N + col] = psq;                                                                                     \
                co_barrier();                                                                                                         \
                if (t == 0) sh.needany[0] = 0u;                                                                                       \
                if (half == act) {                                                                                                    \
                    double vn1 = sh.vn[col];                                                                                          \
                    if (mine) { vn1 = (k + 1 < n) ? sqrt(sh.pdot[col] + sh.pdot[CO_N + col]) : 0.0; sh.vn[col] = vn1; sh.vn[CO_N + col] = vn1; }   \
                    unsigned long long key = live ? co_norm_key(vn1, col) : 0ULL;                                                     \
                    key = wave_max_u64(key);                                                                                          \
                    if (lane == 0) sh.keys[wave & 3] = key;                                                                           \
                }                                                                                                                     \
                co_barrier();                                                                                                         \
                CST_TAIL                                                                    \
            }                                                                                                                         \
        }                                                                                                                             \
    }

#ifdef DQ_QR_STAMPS
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tk0; CST(tk0)
#endif
    CO_BLOCK(0) CO_BLOCK(1) CO_BLOCK(2) CO_BLOCK(3) CO_BLOCK(4) CO_BLOCK(5) CO_BLOCK(6) CO_BLOCK(7)
    CO_BLOCK(8) CO_BLOCK(9) CO_BLOCK(10) CO_BLOCK(11) CO_BLOCK(12) CO_BLOCK(13) CO_BLOCK(14) CO_BLOCK(15)
#undef CO_BLOCK
#ifdef DQ_QR_STAMPS
    if (co_lane() == 0 && (wave == 0 || wave == 5)) {
        unsigned long long tk1; CST(tk1)
        printf("qrc wave %d n=%d total %llu | pivot+publish %llu | B1 %llu | dots(+scalars) %llu | B2 %llu | norms+keys %llu | axpy %llu | Z %llu | recompute %llu\n",
               wave, n, tk1 - tk0, prof[0], prof[1], prof[2], prof[3], prof[4], prof[5], prof[6], prof[7]);
    }
#endif

    // ---- the factored matrix (R0 above each column's pivot position, beta on it, the reflector below), tau, jpvt ----
    CO_IDS
    if (col < n) {
        const int mypos = sh.posc[col];
        const double myscale = sh.vscale[col];
#define CO_STOREV(C) { const int row = 16 * (4 + 2 * (C) + half) + r; if (row < n) A[row + (long)n * col] = row > mypos ? v * myscale : v; }
        CO_RFOREACH_RD_0(CO_STOREV(0)) CO_RFOREACH_RD_1(CO_STOREV(1)) CO_RFOREACH_RD_2(CO_STOREV(2))
        CO_RFOREACH_RD_3(CO_STOREV(3)) CO_RFOREACH_RD_4(CO_STOREV(4)) CO_RFOREACH_RD_5(CO_STOREV(5))
#undef CO_STOREV
#pragma unroll 4
        for (int r = 0; r < 32; ++r) {
            const int row = 16 * (half + 2 * (r >> 4)) + (r & 15);
            if (row < n) { const double v = sh.alds[row * CO_N + col]; A[row + (long)n * col] = row > mypos ? v * myscale : v; }
        }
    }
    if (t < n) { tau[t] = sh.taul[t]; jpvt[t] = sh.jpl[t]; }
}
#undef CO_X

int qr_colown_init_device() {
    DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qrcp_colown_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

int launch_qrcp_colown(Mat A, QrWork w, int n, int n_chains, hipStream_t s) {
    const size_t lds = sizeof(double) * (64 * CO_N + CO_N + 2 * CO_N + CO_N + 2 * CO_N + CO_N + CO_N + 8 + 4) + sizeof(int) * 2 * CO_N + 16 + 2 * CO_N + 64;
    hipLaunchKernelGGL(qrcp_colown_kernel, dim3(1, n_chains), dim3(CO_T), lds, s, A, w, n);
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
