// engine.hip -- the sweep engine (class DQMC of the reference, include/dqmc.h:21-93,
// source/dqmc.cpp) and the C ABI of include/dqmc_hip.h.
//
// An engine owns n_chains independent Markov chains on one GPU; every kernel
// launch advances all of them (blockIdx.y = chain).  All state lives in HBM
// across calls: fields (int8, slice-major), the per-slice exp(+-g*eta) vectors,
// the LDR stack, the current equal-time G and the workspaces.  The host only
// enqueues: a half sweep is one asynchronous stream of launches with no host
// round trip; acceptance decisions, pivoting and error checks happen on device.
#include "common.h"
#include "../../include/dqmc_hip.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace dq {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* get_error() { return g_err.c_str(); }

#define DQ_TRY(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc != 0) return _rc; \
    } while (0)

// hipFuncSetAttribute is per device: every device an engine (or the stateless context) lives on gets the attributes once
int init_device_kernels(int device) {
    static std::mutex mu; static bool done[64] = {};
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0 || device >= 64) { set_error("device ordinal out of range"); return -1; }
    if (done[device]) return 0;
    DQ_HIP(hipSetDevice(device));
    DQ_TRY(update_init_device()); DQ_TRY(update_sm_init_device()); DQ_TRY(qr_init_device()); DQ_TRY(qr_colown_init_device());
    done[device] = true;
    return 0;
}

// ---------------------------------------------------------------------------
// Workspace + stable linear algebra on device (stablelinalg.cpp restated as
// launch sequences).  Shared by the engine and the stateless ABI calls.
// ---------------------------------------------------------------------------
struct LdrRef {                 // a batched LDR triple in HBM
    Mat L; Vec d; Mat R;
    int* jpvt = nullptr;        // [C][n] pivot order storage of this triple (may be null)
    bool* tri = nullptr;        // host flag: R is the permuted-triangular factor of ONE to_LDR (jpvt valid)
};

struct Ctx {
    int n = 0, C = 0, device = 0;
    long nn = 0;
    hipStream_t stream = nullptr;
    static constexpr int NT = 10;              // workspace matrices
    double* pool = nullptr;                    // NT * C * nn
    double* vpool = nullptr;                   // vectors: 8 * C * n
    int* ipool = nullptr;                      // 3 * C * n ints + info
    double* spool = nullptr;                   // scalars: 4 * C
    unsigned long long* qsync = nullptr;       // cooperative QRCP records
    int* qabort = nullptr;                     // cooperative QRCP abort words: C
    double* tinv = nullptr;                    // Gauss-Jordan panel inverses: 2048 * C
    double* qpw = nullptr;                     // panel-pivoted QR (qr_panel.hip): qr_panel_work_doubles(n) * C
    int* qpivpos = nullptr;                    // ... and its pivot positions: n * C
    double* trinv = nullptr;                   // blocked triangular solve: inverses of the 16 x 16 diagonal blocks, 16 * (n + 16) * C
    bool use_tri = false;                      // R^-1 D by tri_solve.hip (n <= 640; the per-column substitution of lu.hip above that)
    bool use_gj = false;                       // n <= 1024, few chains: solves go through lu_gj.hip (DQMC_LU_CLASSIC=1 keeps dgetrf + dgetrs)

    Mat T(int k) const { return Mat{pool + (long)k * C * nn, nn}; }
    Vec V(int k) const { return Vec{vpool + (long)k * C * n, (long)n}; }
    int* jpvt() const { return ipool; }
    int* lperm() const { return ipool + (long)C * n; }
    int* rowpos() const { return ipool + 2L * C * n; }
    int* info() const { return ipool + 3L * C * n; }
    double* logsum() const { return spool; }           // C
    double* scal(int k) const { return spool + (long)k * C; }

    int init(int n_, int C_, int device_) {
        n = n_; C = C_; device = device_; nn = (long)n * n;
        DQ_HIP(hipSetDevice(device));
        DQ_TRY(init_device_kernels(device));
        DQ_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        DQ_HIP(hipMalloc(&pool, sizeof(double) * NT * C * nn));
        DQ_HIP(hipMalloc(&vpool, sizeof(double) * 8 * C * n));
        DQ_HIP(hipMalloc(&ipool, sizeof(int) * (3L * C * n + 4)));
        DQ_HIP(hipMalloc(&spool, sizeof(double) * 4 * C));
        DQ_HIP(hipMalloc(&qsync, sizeof(unsigned long long) * qrcp_coop_sync_granules(n) * C));
        DQ_HIP(hipMalloc(&qabort, sizeof(int) * C));
        DQ_HIP(hipMalloc(&tinv, sizeof(double) * 2048 * C));
        DQ_HIP(hipMalloc(&trinv, sizeof(double) * 16 * (n + 16) * C));
        if (n % 16 == 0 && n <= 1024) {
            DQ_HIP(hipMalloc(&qpw, sizeof(double) * qr_panel_work_doubles(n) * C));
            DQ_HIP(hipMalloc(&qpivpos, sizeof(int) * (size_t)n * C));
        }
        use_tri = n <= 640;     // any number of chains (128 chains at cfg 3: 432.9 against 437.3 ms per step)
        // latency regime only: with many chains per launch the blocked LU + per-column substitution has the higher throughput
        // (128 chains, cfg 3: 458 ms per step against 483 ms with the single-wave panels)
        const int gj_max_chains = getenv("DQMC_GJ_MAX_CHAINS") ? atoi(getenv("DQMC_GJ_MAX_CHAINS")) : 8;
        const int gj_max_n = getenv("DQMC_GJ_MAX_N") ? atoi(getenv("DQMC_GJ_MAX_N")) : 1024;     // test switch (256: blocked LU above); n = 704 ... 1024: 8.4 / 11.7 / 12.0 ms per inv(I + F1 F2) call against 9.9 / 14.0 / 14.1 with the blocked LU
        use_gj = n <= gj_max_n && n <= 1024 && C <= gj_max_chains && getenv("DQMC_LU_CLASSIC") == nullptr;
        DQ_HIP(hipMemsetAsync(ipool, 0, sizeof(int) * (3L * C * n + 4), stream));
        return 0;
    }
    ~Ctx() {
        if (pool) (void)hipFree(pool);
        if (vpool) (void)hipFree(vpool);
        if (ipool) (void)hipFree(ipool);
        if (spool) (void)hipFree(spool);
        if (qsync) (void)hipFree(qsync);
        if (qabort) (void)hipFree(qabort);
        if (tinv) (void)hipFree(tinv);
        if (trinv) (void)hipFree(trinv);
        if (qpw) (void)hipFree(qpw);
        if (qpivpos) (void)hipFree(qpivpos);
        if (stream) (void)hipStreamDestroy(stream);
    }

    int gemm(CMat A, CMat B, Mat Cm, CVec rs = CVec(), CVec ks = CVec(), CVec cs = CVec(), int transA = 0, int accumulate = 0, Mat CT = Mat{nullptr, 0}) {
        GemmDesc g; g.A = A; g.B = B; g.C = Cm; g.CT = CT; g.rs = rs; g.ks = ks; g.cs = cs; g.n = n; g.transA = transA; g.accumulate = accumulate;
        return launch_gemm(g, C, stream);
    }
    // stablelinalg::to_LDR (source/stablelinalg.cpp:35-55); A is destroyed
    // `direct`: out.R receives the QR's own R factor (not a product), so out can take the
    // triangular fast path later; its pivot order is written straight into out.jpvt.
    int to_ldr(Mat A, LdrRef out, bool direct = true) {
        const bool keep = direct && out.jpvt != nullptr;
        QrWork w{V(7).p, (long)n, keep ? out.jpvt : jpvt(), (long)n};
        w.sync = qsync; w.sync_stride = qrcp_coop_sync_granules(n); w.abort_words = qabort; w.info = info();
        w.pw = qpw; w.pw_stride = qr_panel_work_doubles(n); w.pivpos = qpivpos; w.pivpos_stride = n;
        if (out.tri) *out.tri = keep;
        return launch_to_ldr(A, out.L, out.d, out.R, w, n, C, stream);
    }
    // X = F.R^-1 diag(dinv): permuted triangular solve when F.R is a single QR factor, LU otherwise
    int r_inverse_scaled(LdrRef F, CVec dinv, Mat X, Mat lu_scratch) {
        if (F.tri && *F.tri && F.jpvt) {
            if (use_tri) return launch_tri_solve(F.R, F.jpvt, n, X, dinv, trinv, 16L * (n + 16), n, C, stream);
            return launch_lu_solve(F.R, F.jpvt, n, X, dinv, 2, n, C, stream);
        }
        DQ_TRY(launch_copy(F.R, lu_scratch, nn, C, stream));
        DQ_TRY(launch_lu(lu_scratch, lperm(), n, nullptr, 0, info(), n, C, stream, rowpos(), n));
        return launch_lu_solve(lu_scratch, lperm(), n, X, dinv, 1, n, C, stream);
    }
    // stablelinalg::mat_mul_ldr (source/stablelinalg.cpp:69-79): out = M * F   (uses T0,T1)
    int mat_mul_ldr(CMat M, LdrRef F, LdrRef out) {
        DQ_TRY(gemm(M, F.L, T(0), CVec(), CVec(), CVec(F.d)));           // (M L) diag(d)
        LdrRef q{out.L, out.d, T(1)};
        DQ_TRY(to_ldr(T(0), q));
        if (out.tri) *out.tri = false;                                   // R becomes a product
        return gemm(T(1), F.R, out.R);                                   // r * R
    }
    // stablelinalg::ldr_mul_mat (source/stablelinalg.cpp:57-67): out = F * M   (uses T0,T1)
    int ldr_mul_mat(LdrRef F, CMat M, LdrRef out) {
        DQ_TRY(gemm(F.R, M, T(0), CVec(F.d)));                            // diag(d) (R M)
        LdrRef q{T(1), out.d, out.R, out.jpvt, out.tri};                  // R = the QR's own factor
        DQ_TRY(to_ldr(T(0), q));
        return gemm(F.L, T(1), out.L);                                   // L * q
    }
    // stablelinalg::ldr_mul_ldr (source/stablelinalg.cpp:81-92): out = F1 * F2   (uses T0,T1,T2)
    int ldr_mul_ldr(LdrRef F1, LdrRef F2, LdrRef out) {
        DQ_TRY(gemm(F1.R, F2.L, T(0), CVec(F1.d), CVec(), CVec(F2.d)));    // diag(d1) (R1 L2) diag(d2)
        LdrRef q{T(1), out.d, T(2)};
        DQ_TRY(to_ldr(T(0), q));
        if (out.tri) *out.tri = false;
        DQ_TRY(gemm(F1.L, T(1), out.L));
        return gemm(T(2), F2.R, out.R);
    }
    // Y = M^-1 RHS (arma::solve): blocked Gauss-Jordan for n <= 256 (result in T9's neighbour `out`), dgetrf + dgetrs otherwise
    // (result overwrites RHS).  Returns the matrix holding Y in *Y.  M and RHS are destroyed.  (uses T9 as scratch)
    int solve(Mat M, Mat RHS, Mat out, double* logdet_acc, Mat* Y) {
        if (use_gj) {
            DQ_TRY(launch_gj_solve(M, RHS, out, T(9), tinv, lperm(), n, rowpos(), n, logdet_acc, 1, info(), n, C, stream));
            *Y = out; return 0;
        }
        DQ_TRY(launch_lu(M, lperm(), n, logdet_acc, 1, info(), n, C, stream, rowpos(), n));
        DQ_TRY(launch_lu_solve(M, lperm(), n, RHS, CVec(), 0, n, C, stream));
        *Y = RHS; return 0;
    }
    // stablelinalg::inv_I_plus_ldr (source/stablelinalg.cpp:94-126)   (uses T0..T4, V0,V1)
    // G = X M^-1 is evaluated as the reference does, through the transposed system M^T G^T = X^T;
    // log|det M| comes from that same factorisation (det M^T = det M).
    int inv_I_plus_ldr(LdrRef F, Mat G, double* logdet /*device, C*/) {
        DQ_TRY(launch_split_d(F.d, V(0), V(1), logdet, n, C, stream));                 // V0 = 1/Dl, V1 = Ds, logdet = sum log Dl
        DQ_TRY(r_inverse_scaled(F, V(0), T(1), T(0)));                                  // X = R^-1 diag(1/Dl)
        DQ_TRY(launch_add_scaled_cols(T(1), F.L, V(1), T(2), n, C, stream));            // M = X + L diag(Ds)
        DQ_TRY(launch_transpose_scale(T(2), T(3), CVec(), n, C, stream));               // M^T
        DQ_TRY(launch_transpose_scale(T(1), T(4), CVec(), n, C, stream));               // X^T
        Mat Y; DQ_TRY(solve(T(3), T(4), T(0), logdet, &Y));                             // G^T; logdet += log|det M|
        return launch_transpose_scale(Y, G, CVec(), n, C, stream);
    }
    // stablelinalg::inv_I_plus_ldr_mul_ldr (source/stablelinalg.cpp:128-158)   (uses T0..T3, V0..V3)
    int inv_I_plus_ldr_mul_ldr(LdrRef F1, LdrRef F2, Mat G) {
        DQ_TRY(launch_split_d2(F1.d, V(0), V(1), F2.d, V(2), V(3), n, C, stream));     // 1/D1l, D1s | 1/D2l, D2s
        DQ_TRY(r_inverse_scaled(F2, V(2), T(1), T(0)));                                 // X = R2^-1 diag(1/D2l)
        DQ_TRY(gemm(F1.L, T(1), T(2), V(0), CVec(), CVec(), 1));                        // TermA = diag(1/D1l) L1^T X
        DQ_TRY(gemm(F1.R, F2.L, T(2), V(1), CVec(), V(3), 0, 1));                       // M = TermA + diag(D1s) R1 L2 diag(D2s)
        DQ_TRY(launch_transpose_scale(F1.L, T(3), V(0), n, C, stream));                 // RHS = diag(1/D1l) L1^T
        Mat Y; DQ_TRY(solve(T(2), T(3), T(0), nullptr, &Y));                            // Y = M^-1 RHS
        return gemm(T(1), Y, G);                                                        // G = X Y
    }
    // stablelinalg::inv_invldr_plus_ldr (source/stablelinalg.cpp:160-190): G = [F1^-1 + F2]^-1 (negated on request)   (uses T0..T3, V0..V3)
    int inv_invldr_plus_ldr(LdrRef F1, LdrRef F2, Mat G, bool negate) {
        DQ_TRY(launch_split_d(F1.d, V(0), V(1), nullptr, n, C, stream));               // 1/D1l, D1s
        DQ_TRY(launch_split_d(F2.d, V(2), V(3), nullptr, n, C, stream));               // 1/D2l, D2s
        DQ_TRY(r_inverse_scaled(F2, V(2), T(1), T(0)));                                 // X = R2^-1 diag(1/D2l)
        DQ_TRY(gemm(F1.L, T(1), T(2), V(0), CVec(), CVec(), 1));                        // TermA = diag(1/D1l) L1^T X
        DQ_TRY(gemm(F1.R, F2.L, T(2), V(1), CVec(), V(3), 0, 1));                       // M = TermA + diag(D1s) R1 L2 diag(D2s)
        DQ_TRY(launch_scale_rows(F1.R, V(1), T(3), n, C, stream));                      // RHS = diag(D1s) R1
        Mat Y; DQ_TRY(solve(T(2), T(3), T(0), nullptr, &Y));                            // Y = M^-1 RHS
        DQ_TRY(gemm(T(1), Y, G));                                                       // G = X Y
        if (negate) DQ_TRY(launch_axpb_identity(G, G, -1.0, 0.0, n, C, stream));
        return 0;
    }
};

// ---------------------------------------------------------------------------
// Engine
// ---------------------------------------------------------------------------
struct Engine {
    int n = 0, nt = 0, n_stab = 0, n_stack = 0, C = 1, device = 0;
    long nn = 0;
    std::vector<int> loc_l_end;
    std::vector<double> g_host, gamma_host, eta_host;
    std::vector<unsigned char> seen;                             // scratch of the random-stream validation
    Ctx ctx;
    hipStream_t s = nullptr;

    double* expK = nullptr; double* invexpK = nullptr;           // [C][nn]
    double* expKh = nullptr; double* invexpKh = nullptr;         // [nn] each: exp(-+ dtau K / 2) of dqmc_half_warp, uploaded on first use
    double* hwOut = nullptr;                                     // [C][nn] result of dqmc_half_warp
    // checkerboard break-up of exp(-+dtau K) (dqmc_set_checkerboard): wraps and B-bar products apply the pair factors directly
    bool cb = false; int cb_groups = 0;
    int* cb_partner = nullptr;                                   // [cb_groups][n]
    double* cb_par = nullptr;                                    // [C][4] cosh, sinh, f, 1/f
    int8_t* fields = nullptr;                                    // [C][nt][n]
    double* expv = nullptr; double* invexpv = nullptr;           // [C][nt][n]
    UpdateTables* tabs = nullptr; double* tab8 = nullptr;        // [C], [C][8]
    bool gt_valid = false;                                       // GT == G^T right now (set by the wraps, cleared by everything else that writes G)
    double* G = nullptr; double* Gtmp = nullptr; double* GT = nullptr;   // [C][nn]; GT: transposed copy for the local-update walk
    double* pg_eye = nullptr; double* pg_ones = nullptr;         // identity [nn] and ones [n]: operands of the first piggybacked B-bar factor of a block (single chain)
    double* bb0 = nullptr; double* bb1 = nullptr;                // Bbar ping-pong
    double* stackL = nullptr; double* stackD = nullptr; double* stackR = nullptr;
    double* tmpL = nullptr; double* tmpD = nullptr; double* tmpR = nullptr;   // one spare LDR (init_stacks)
    int* stackP = nullptr;                                       // [n_stack][C][n] pivot order of each entry's R
    std::unique_ptr<bool[]> stack_tri;                           // [n_stack] R is a single permuted-triangular factor
    double* logdet = nullptr;                                    // [C]
    int32_t* rs_perm = nullptr; uint8_t* rs_k = nullptr; double* rs_u = nullptr;   // [C][nt][n]
    void* h_stage = nullptr; size_t h_stage_bytes = 0;           // pinned staging for the random stream
    hipEvent_t stage_free = nullptr;
    double* Upanel = nullptr; double* Wpanel = nullptr;          // [C][KD][n]
    double* Cpanel = nullptr;                                    // [C][KD][KD]
    double* ibuf = nullptr; int* ijp = nullptr;                  // batched initialisation (init_batched): Bbar ping-pong, L, R, d, tau / pivots of every block
    int* state = nullptr;                                        // [C][4]
    double* prep = nullptr;                                      // [C][4n]
    double* meas_now = nullptr; double* meas_sum = nullptr;      // [C][3 + n] equal-time observables: last evaluation / bin sums
    // unequal-time path (allocated by the first sweep_unequal): Gtt / Gt0 / G0t [nt + 1][C][nn], B(tau,0) ping-pong LDRs, scratch
    double* utG[3] = {nullptr, nullptr, nullptr}; double* utTmp = nullptr; double* utErr = nullptr;
    double* utL[2] = {nullptr, nullptr}; double* utD[2] = {nullptr, nullptr}; double* utR[2] = {nullptr, nullptr}; int* utP[2] = {nullptr, nullptr};
    bool utTri[2] = {false, false}; bool ut_valid = false;
    double* utMeasNow = nullptr; double* utMeasSum = nullptr; long long ut_meas_count = 0;     // [C][3][nt + 1][n] dynamical observables: last / bin sums
    long long meas_count = 0;                                    // measurements accumulated in meas_sum
    char* slice_sync = nullptr;                                  // [C][SLICE_SYNC_BYTES = 2 KiB] hand-off words of the persistent slice kernels (SliceSync, common.h)
    bool persistent = false;                                     // holds a CU reservation for the single-launch slice kernel (slice_reserve)
    bool handoff_failed = false;                                 // a hand-off of a persistent kernel timed out once: this engine stays on the kernel pairs from then on
    unsigned slice_epoch = 0;                                    // launches of the persistent slice kernel so far: the tag of its hand-off words (SliceSync, common.h)
    int slice_absent_l = -1;                                     // ...=<tile>:<slice>: only in the launch of that time slice
    int slice_late_tile = -1, slice_late_us = 0;                 // DQMC_DEBUG_SLICE_LATE=<tile>:<us>: that flush workgroup checks in only after <us> microseconds (test of a LATE arrival)
    int slice_absent_tile = -1;                                  // DQMC_DEBUG_SLICE_ABSENT=<tile>, read when the engine is created: that flush workgroup never checks in (test of the solo fall-back)
    int* acc = nullptr;                                          // [C][nt]
    double* err = nullptr;                                       // [C][n_stack]
    DevStats* dstats = nullptr;                                  // [C]
    double* r1scratch = nullptr;
    bool stack_valid = false;
    // profiling of the local-update kernels
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pairs;
    size_t ev_used = 0;
    double upd_ms = 0.0; long long upd_launches = 0; long long upd_accept_base = 0;

    Mat mG() const { return Mat{G, nn}; }
    LdrRef stk(int i) const {
        return LdrRef{Mat{stackL + (long)i * C * nn, nn}, Vec{stackD + (long)i * C * n, (long)n}, Mat{stackR + (long)i * C * nn, nn},
                      stackP + (long)i * C * n, &stack_tri[i]};
    }
    CVec ev(int l) const { return CVec(expv + (long)l * n, (long)nt * n); }
    CVec iev(int l) const { return CVec(invexpv + (long)l * n, (long)nt * n); }
    int stack_idx(int l) const { return l / n_stab; }            // include/dqmc.h:47
    int local_l(int l) const { return l % n_stab; }              // include/dqmc.h:48

    ~Engine() {
        if (s) (void)hipStreamSynchronize(s);
        if (persistent) slice_release(device, n, C);
        for (auto& p : ev_pairs) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        if (stage_free) (void)hipEventDestroy(stage_free);
        void* ptrs[] = {expKh, invexpKh, hwOut, utMeasNow, utMeasSum, utG[0], utG[1], utG[2], utTmp, utErr, utL[0], utL[1], utD[0], utD[1], utR[0], utR[1], utP[0], utP[1], meas_now, meas_sum, GT, slice_sync, prep, stackP, expK, invexpK, cb_partner, cb_par, fields, expv, invexpv, tabs, tab8, G, pg_eye, pg_ones, Gtmp, bb0, bb1, stackL, stackD, stackR, tmpL, tmpD, tmpR,
                        logdet, rs_perm, rs_k, rs_u, Upanel, Wpanel, Cpanel, ibuf, ijp, state, acc, err, dstats, r1scratch};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        if (h_stage) (void)hipHostFree(h_stage);
    }

    template <class T> int dalloc(T** p, size_t count) { DQ_HIP(hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * count)); return 0; }

    int create(int device_, int C_, int n_, int nt_, int n_stab_, const double* g, const double* gamma, const double* eta,
               const double* eK, const double* ieK) {
        device = device_; C = C_; n = n_; nt = nt_; n_stab = n_stab_; nn = (long)n * n;
        n_stack = (int)std::ceil(static_cast<double>(nt) / n_stab);                       // source/dqmc.cpp:10
        loc_l_end.assign(n_stack, n_stab - 1);
        if (nt % n_stab != 0) loc_l_end[n_stack - 1] = nt % n_stab - 1;                    // source/dqmc.cpp:13-18
        g_host.assign(g, g + C); gamma_host.assign(gamma, gamma + 4); eta_host.assign(eta, eta + 4);
        DQ_TRY(ctx.init(n, C, device)); s = ctx.stream;
        persistent = slice_reserve(device, n, C);
        if (const char* a = getenv("DQMC_DEBUG_SLICE_ABSENT")) { slice_absent_tile = atoi(a); if (const char* c = strchr(a, ':')) slice_absent_l = atoi(c + 1); }
        if (const char* a = getenv("DQMC_DEBUG_SLICE_LATE")) { slice_late_tile = atoi(a); if (const char* c = strchr(a, ':')) slice_late_us = atoi(c + 1); }
        DQ_TRY(dalloc(&expK, C * nn)); DQ_TRY(dalloc(&invexpK, C * nn));
        DQ_TRY(dalloc(&fields, (size_t)C * nt * n)); DQ_TRY(dalloc(&expv, (size_t)C * nt * n)); DQ_TRY(dalloc(&invexpv, (size_t)C * nt * n));
        DQ_TRY(dalloc(&tabs, C)); DQ_TRY(dalloc(&tab8, (size_t)C * 8));
        if (C == 1) { DQ_TRY(dalloc(&pg_eye, nn)); DQ_TRY(dalloc(&pg_ones, (size_t)n)); DQ_TRY(launch_set_identity(Mat{pg_eye, nn}, n, 1, s)); const std::vector<double> one_h((size_t)n, 1.0); DQ_HIP(hipMemcpy(pg_ones, one_h.data(), sizeof(double) * n, hipMemcpyHostToDevice)); }
        DQ_TRY(dalloc(&G, C * nn)); DQ_TRY(dalloc(&Gtmp, C * nn)); DQ_TRY(dalloc(&GT, C * nn)); DQ_TRY(dalloc(&bb0, C * nn)); DQ_TRY(dalloc(&bb1, C * nn));
        DQ_TRY(dalloc(&stackL, (size_t)n_stack * C * nn)); DQ_TRY(dalloc(&stackD, (size_t)n_stack * C * n)); DQ_TRY(dalloc(&stackR, (size_t)n_stack * C * nn));
        DQ_TRY(dalloc(&stackP, (size_t)n_stack * C * n)); stack_tri.reset(new bool[n_stack]());
        DQ_TRY(dalloc(&tmpL, C * nn)); DQ_TRY(dalloc(&tmpD, (size_t)C * n)); DQ_TRY(dalloc(&tmpR, C * nn));
        DQ_TRY(dalloc(&logdet, C));
        DQ_TRY(dalloc(&rs_perm, (size_t)C * nt * n)); DQ_TRY(dalloc(&rs_k, (size_t)C * nt * n)); DQ_TRY(dalloc(&rs_u, (size_t)C * nt * n));
        DQ_TRY(dalloc(&Upanel, (size_t)C * UPDATE_KD * n)); DQ_TRY(dalloc(&Wpanel, (size_t)C * UPDATE_KD * n)); DQ_TRY(dalloc(&Cpanel, (size_t)C * UPDATE_KD * UPDATE_KD));
        DQ_TRY(dalloc(&state, (size_t)C * 4)); DQ_TRY(dalloc(&prep, (size_t)C * 4 * n)); DQ_TRY(dalloc(&meas_now, (size_t)C * (3 + n))); DQ_TRY(dalloc(&meas_sum, (size_t)C * (3 + n))); DQ_HIP(hipMemsetAsync(meas_sum, 0, sizeof(double) * C * (3 + n), s)); DQ_TRY(dalloc(&slice_sync, (size_t)C * SLICE_SYNC_BYTES)); DQ_HIP(hipMemsetAsync(slice_sync, 0, (size_t)C * SLICE_SYNC_BYTES, s)); DQ_TRY(dalloc(&acc, (size_t)C * nt)); DQ_TRY(dalloc(&err, (size_t)C * n_stack));
        DQ_TRY(dalloc(&dstats, C)); DQ_TRY(dalloc(&r1scratch, (size_t)C * (2 * n + 1)));
        h_stage_bytes = (size_t)C * nt * n * (sizeof(int32_t) + sizeof(uint8_t) + sizeof(double));
        DQ_HIP(hipHostMalloc(&h_stage, h_stage_bytes, hipHostMallocDefault));
        DQ_HIP(hipEventCreateWithFlags(&stage_free, hipEventDisableTiming));
        DQ_HIP(hipEventRecord(stage_free, s));
        DQ_HIP(hipMemcpyAsync(expK, eK, sizeof(double) * C * nn, hipMemcpyHostToDevice, s));
        DQ_HIP(hipMemcpyAsync(invexpK, ieK, sizeof(double) * C * nn, hipMemcpyHostToDevice, s));
        DQ_HIP(hipMemsetAsync(fields, 0, (size_t)C * nt * n, s));
        DQ_HIP(hipMemsetAsync(G, 0, sizeof(double) * C * nn, s));
        DQ_HIP(hipMemsetAsync(dstats, 0, sizeof(DevStats) * C, s));
        DQ_HIP(hipMemsetAsync(logdet, 0, sizeof(double) * C, s));
        DQ_HIP(hipMemsetAsync(state, 0, sizeof(int) * C * 4, s));
        // model tables, same expressions as source/model.cpp:99-122, :62-84
        std::vector<UpdateTables> ht(C); std::vector<double> h8((size_t)C * 8);
        static const int proposal[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
        const double alpha = -1.0;
        for (int c = 0; c < C; ++c) {
            std::memset(&ht[c], 0, sizeof(UpdateTables));
            for (int o = 0; o < 4; ++o) for (int k = 0; k < 3; ++k) {
                const int nf = proposal[o][k];
                const double gammaR = gamma[nf] / gamma[o];
                const double d_eta = eta[nf] - eta[o];
                const double bosonR = std::exp(alpha * g[c] * d_eta);
                ht[c].rb[o][k] = gammaR * bosonR;
                ht[c].delta[o][k] = (1.0 / bosonR) - 1.0;
            }
            for (int f = 0; f < 4; ++f) { ht[c].ev[f] = std::exp(g[c] * eta[f]); ht[c].iev[f] = std::exp(-g[c] * eta[f]); h8[(size_t)c * 8 + f] = ht[c].ev[f]; h8[(size_t)c * 8 + 4 + f] = ht[c].iev[f]; }
        }
        DQ_HIP(hipMemcpyAsync(tabs, ht.data(), sizeof(UpdateTables) * C, hipMemcpyHostToDevice, s));
        DQ_HIP(hipMemcpyAsync(tab8, h8.data(), sizeof(double) * C * 8, hipMemcpyHostToDevice, s));
        DQ_TRY(launch_build_expv(fields, (long)nt * n, nt, n, tab8, expv, invexpv, (long)nt * n, C, s));
        DQ_HIP(hipStreamSynchronize(s));
        return 0;
    }

    // one launch of the checkerboard kernel on this engine's pair tables (chains / par_stride differ for the batched initialisation)
    int cb_apply(CMat in, Mat out, Mat outT, bool reverse, bool inverse, CVec rs_in, CVec cs_in, CVec rs_out, CVec cs_out, int chains, long par_stride) {
        CbDesc d; d.in = in; d.out = out; d.outT = outT; d.partner = cb_partner; d.n_groups = cb_groups; d.reverse = reverse ? 1 : 0; d.inverse = inverse ? 1 : 0;
        d.par = cb_par; d.par_stride = par_stride; d.rs_in = rs_in; d.cs_in = cs_in; d.rs_out = rs_out; d.cs_out = cs_out; d.n = n;
        return launch_cb_apply(d, chains, s);
    }
    // dqmc_set_checkerboard: pair tables to the device, then E and E^-1 as dense matrices (the kernel applied to I) for the
    // paths that stay GEMMs (the unequal-time series and the first factor of a B-bar product)
    int set_checkerboard(int n_groups, const int32_t* bonds, const int32_t* group_sizes, const double* ch, const double* sh, const double* f) {
        if (n_groups < 1 || n_groups > 64 || !bonds || !group_sizes || !ch || !sh || !f) { set_error("set_checkerboard: bad argument"); return DQMC_EINVAL; }
        if (n > 4096) { set_error("set_checkerboard: n_sites > 4096"); return DQMC_EINVAL; }
        std::vector<int> partner((size_t)n_groups * n);
        size_t b = 0;
        for (int g = 0; g < n_groups; ++g) {
            int* pt = partner.data() + (size_t)g * n;
            for (int i = 0; i < n; ++i) pt[i] = i;
            if (group_sizes[g] < 0 || group_sizes[g] > n / 2) { set_error("set_checkerboard: a group holds at most n_sites/2 bonds"); return DQMC_EINVAL; }
            for (int k = 0; k < group_sizes[g]; ++k, ++b) {
                const int i = bonds[2 * b], j = bonds[2 * b + 1];
                if (i < 0 || i >= n || j < 0 || j >= n || i == j) { set_error("set_checkerboard: bond site out of range"); return DQMC_EINVAL; }
                if (pt[i] != i || pt[j] != j) { set_error("set_checkerboard: the bonds of one group must be disjoint"); return DQMC_EINVAL; }
                pt[i] = j; pt[j] = i;
            }
        }
        std::vector<double> par((size_t)C * 4);
        for (int c = 0; c < C; ++c) {
            if (!(f[c] > 0.0) || !std::isfinite(ch[c]) || !std::isfinite(sh[c])) { set_error("set_checkerboard: diag_factor must be positive, cosh / sinh finite"); return DQMC_EINVAL; }
            par[4 * c] = ch[c]; par[4 * c + 1] = sh[c]; par[4 * c + 2] = f[c]; par[4 * c + 3] = 1.0 / f[c];
        }
        DQ_HIP(hipStreamSynchronize(s));
        if (cb_partner) { (void)hipFree(cb_partner); cb_partner = nullptr; }
        DQ_TRY(dalloc(&cb_partner, partner.size()));
        if (!cb_par) DQ_TRY(dalloc(&cb_par, (size_t)C * 4));
        DQ_HIP(hipMemcpy(cb_partner, partner.data(), sizeof(int) * partner.size(), hipMemcpyHostToDevice));
        DQ_HIP(hipMemcpy(cb_par, par.data(), sizeof(double) * par.size(), hipMemcpyHostToDevice));
        cb_groups = n_groups; cb = true;
        Mat id{bb0, nn};
        DQ_TRY(launch_set_identity(id, n, C, s));
        DQ_TRY(cb_apply(id, Mat{expK, nn}, Mat{nullptr, 0}, false, false, CVec(), CVec(), CVec(), CVec(), C, 4));      // E = f E_{G-1} ... E_0
        DQ_TRY(cb_apply(id, Mat{invexpK, nn}, Mat{nullptr, 0}, true, true, CVec(), CVec(), CVec(), CVec(), C, 4));     // E^-1 = E_0^-1 ... E_{G-1}^-1 / f
        stack_valid = false; gt_valid = false; ut_valid = false;
        DQ_HIP(hipStreamSynchronize(s));
        return 0;
    }

    // DQMC::calculate_Bbar (source/dqmc.cpp:88-105) without the multiply by I: result in *out
    int Bbar(int is, Mat* out) {
        const int l0 = is * n_stab;
        Mat cur{bb0, nn}, nxt{bb1, nn};
        DQ_TRY(launch_scale_rows(CMat(expK, nn), ev(l0), cur, n, C, s));                  // B_l0 = diag(expV) expK
        for (int loc = 1; loc <= loc_l_end[is]; ++loc) {
            if (cb) DQ_TRY(cb_apply(cur, nxt, Mat{nullptr, 0}, false, false, CVec(), CVec(), ev(l0 + loc), CVec(), C, 4));
            else DQ_TRY(ctx.gemm(CMat(expK, nn), cur, nxt, ev(l0 + loc)));                // B_l * Bbar
            std::swap(cur, nxt);
        }
        *out = cur; return 0;
    }
    // DQMC::init_stacks + init_greenfunctions (source/dqmc.cpp:43-72)
    int init() {
        // Single chain, N <= 256, whole blocks: the n_stack products Bbar_i and their to_LDR factorisations do not depend on each
        // other (source/dqmc.cpp:47-53 computes them one after the other), so they run as ONE batch with the block index in the
        // kernels' chain dimension -- n_stab - 1 launches of the 64x64-tile GEMM for all blocks, one QRCP launch with a workgroup
        // per block on its own CU (20 x 0.84 ms side by side instead of in a row) -- and only the n_stack - 1 ldr_mul_ldr products,
        // which are a chain, stay sequential.  Halves the cost of an initialisation, i.e. of a replica-exchange round.
        if (C == 1 && n <= 256 && nt % n_stab == 0 && n_stack >= 2) return init_batched();
        LdrRef tmp{Mat{tmpL, nn}, Vec{tmpD, (long)n}, Mat{tmpR, nn}};
        for (int i = n_stack - 1; i >= 0; --i) {
            Mat bb; DQ_TRY(Bbar(i, &bb));
            if (i == n_stack - 1) DQ_TRY(ctx.to_ldr(bb, stk(i)));
            else { DQ_TRY(ctx.to_ldr(bb, tmp)); DQ_TRY(ctx.ldr_mul_ldr(stk(i + 1), tmp, stk(i))); }
        }
        stack_valid = true;
        gt_valid = false;
        return ctx.inv_I_plus_ldr(stk(0), mG(), logdet);
    }
    int init_batched() {
        const int S = n_stack;
        if (!ibuf) {
            DQ_TRY(dalloc(&ibuf, (size_t)4 * S * nn + (size_t)2 * S * n));
            DQ_TRY(dalloc(&ijp, (size_t)S * n));
        }
        double* ib0 = ibuf; double* ib1 = ib0 + (size_t)S * nn; double* iL = ib1 + (size_t)S * nn; double* iR = iL + (size_t)S * nn;
        double* iD = iR + (size_t)S * nn; double* iTau = iD + (size_t)S * n;
        const long bstride = (long)n_stab * n;                                            // exp(V) of block i starts n_stab slices further
        Mat cur{ib0, nn}, nxt{ib1, nn};
        DQ_TRY(launch_scale_rows(CMat(expK, 0), CVec(expv, bstride), cur, n, S, s));        // B_l0 of every block
        for (int loc = 1; loc < n_stab; ++loc) {
            if (cb) DQ_TRY(cb_apply(cur, nxt, Mat{nullptr, 0}, false, false, CVec(), CVec(), CVec(expv + (long)loc * n, bstride), CVec(), S, 0));
            else {
                GemmDesc g; g.A = CMat(expK, 0); g.B = cur; g.C = nxt; g.rs = CVec(expv + (long)loc * n, bstride); g.n = n;
                DQ_TRY(launch_gemm(g, S, s));                                             // B_l * Bbar, all blocks
            }
            std::swap(cur, nxt);
        }
        QrWork w{iTau, (long)n, ijp, (long)n};
        w.info = ctx.info();
        DQ_TRY(launch_to_ldr(cur, Mat{iL, nn}, Vec{iD, (long)n}, Mat{iR, nn}, w, n, S, s));   // to_LDR(Bbar_i) for every block
        // stack[S - 1] = its own factorisation (R a single permuted-triangular factor), then the chain of products
        LdrRef last = stk(S - 1);
        DQ_TRY(launch_copy(CMat(iL + (size_t)(S - 1) * nn, nn), last.L, nn, 1, s));
        DQ_TRY(launch_copy(CMat(iR + (size_t)(S - 1) * nn, nn), last.R, nn, 1, s));
        DQ_HIP(hipMemcpyAsync(last.d.p, iD + (size_t)(S - 1) * n, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
        DQ_HIP(hipMemcpyAsync(last.jpvt, ijp + (size_t)(S - 1) * n, sizeof(int) * n, hipMemcpyDeviceToDevice, s));
        *last.tri = true;
        for (int i = S - 2; i >= 0; --i) {
            LdrRef tmp{Mat{iL + (size_t)i * nn, nn}, Vec{iD + (size_t)i * n, (long)n}, Mat{iR + (size_t)i * nn, nn}};
            DQ_TRY(ctx.ldr_mul_ldr(stk(i + 1), tmp, stk(i)));
        }
        stack_valid = true;
        gt_valid = false;
        return ctx.inv_I_plus_ldr(stk(0), mG(), logdet);
    }
    // DQMC::propagate_GF_forward (source/dqmc.cpp:113-132): G = B_l G B_l^-1
    int wrap_forward(int l) {
        if (cb) {
            // T^T = (E G)^T;  (T E^-1)^T = E^-T T^T with E^-T = E_{G-1}^-1 ... E_0^-1 / f: groups 0 .. G-1, inverse;  G' = diag(ev) (T E^-1) diag(iev)
            DQ_TRY(cb_apply(mG(), Mat{nullptr, 0}, ctx.T(0), false, false, CVec(), CVec(), CVec(), CVec(), C, 4));
            gt_valid = use_gt();
            return cb_apply(ctx.T(0), Mat{GT, nn}, mG(), false, true, CVec(), CVec(), iev(l), ev(l), C, 4);
        }
        DQ_TRY(ctx.gemm(CMat(expK, nn), mG(), ctx.T(0)));
        gt_valid = use_gt();                                          // the GEMM that writes G writes GT as well
        return ctx.gemm(ctx.T(0), CMat(invexpK, nn), mG(), ev(l), CVec(), iev(l), 0, 0, use_gt() ? Mat{GT, nn} : Mat{nullptr, 0});
    }
    // The B-bar chain rides on the wraps (single chain, dense kinetic factor, N <= 256: the regime where a 256^3 product IS its launch).
    // calculate_Bbar (source/dqmc.cpp:88-105) is n_stab - 1 dependent products per stabilisation, 5.6 us each of which ~4.5 us are the launch:
    // each of them has an operand in common with a wrap product of the same sweep, so it is computed as a SECOND "chain" of that launch
    // (the GEMM kernels take the chain index in blockIdx.y and per-operand chain strides: chain 1's operands are simply other buffers).
    //   forward  (wrap BEFORE the update of its slice): the wrap of slice l starts with expK * G; beside it P <- diag(e^{V_{l-1}}) expK P
    //            with the fields slice l - 1 has just been given -- the factor of the block's last slice is one explicit product;
    //   backward (wrap AFTER the update): the wrap of slice l ends with T diag(e^{V_l}) expK; beside it P <- P diag(e^{V_l}) expK = P B_l.
    // The first factor of a block multiplies the identity.  Same products, same kernel, same association in the forward sweep as Bbar();
    // in the backward sweep the chain is associated from the other end.
    bool piggyback() const { return C == 1 && !cb && n <= 256 && pg_eye != nullptr; }
    int wrap_forward_piggy(int l, const double* Pprev, double* Pnext) {
        GemmDesc g; g.A = CMat(expK, 0); g.B = CMat(G, (long)(Pprev - G)); g.C = Mat{ctx.T(0).p, (long)(Pnext - ctx.T(0).p)};
        g.rs = CVec(pg_ones, (long)((expv + (long)(l - 1) * n) - pg_ones)); g.n = n;
        DQ_TRY(launch_gemm(g, 2, s));
        gt_valid = use_gt();
        return ctx.gemm(ctx.T(0), CMat(invexpK, nn), mG(), ev(l), CVec(), iev(l), 0, 0, use_gt() ? Mat{GT, nn} : Mat{nullptr, 0});
    }
    int wrap_backward_piggy(int l, const double* Pprev, double* Pnext) {
        DQ_TRY(ctx.gemm(CMat(invexpK, nn), mG(), ctx.T(0), CVec(), iev(l)));
        gt_valid = use_gt();
        GemmDesc g; g.A = CMat(ctx.T(0).p, (long)(Pprev - ctx.T(0).p)); g.B = CMat(expK, 0); g.C = Mat{G, (long)(Pnext - G)};
        g.ks = CVec(expv + (long)l * n, 0); g.n = n;
        if (use_gt()) g.CT = Mat{GT, (long)(ctx.T(1).p - GT)};          // chain 1's transposed copy goes to scratch
        return launch_gemm(g, 2, s);
    }
    // DQMC::propagate_GF_backward (source/dqmc.cpp:169-187): G = B_l^-1 G B_l
    int wrap_backward(int l) {
        if (cb) {
            // T^T = (E^-1 diag(iev) G diag(ev))^T;  (T E)^T = E^T T^T with E^T = f E_0 ... E_{G-1}: groups G-1 .. 0, forward
            DQ_TRY(cb_apply(mG(), Mat{nullptr, 0}, ctx.T(0), true, true, iev(l), ev(l), CVec(), CVec(), C, 4));
            gt_valid = use_gt();
            return cb_apply(ctx.T(0), Mat{GT, nn}, mG(), true, false, CVec(), CVec(), CVec(), CVec(), C, 4);
        }
        DQ_TRY(ctx.gemm(CMat(invexpK, nn), mG(), ctx.T(0), CVec(), iev(l)));
        gt_valid = use_gt();
        return ctx.gemm(ctx.T(0), CMat(expK, nn), mG(), CVec(), ev(l), CVec(), 0, 0, use_gt() ? Mat{GT, nn} : Mat{nullptr, 0});
    }
    // the walk reads rows of G from a transposed copy: the register walk (n <= 256) and the persistent sub-matrix kernel (any n)
    bool use_gt() const { return n <= 256 || (persistent && !handoff_failed); }     // n > 256: only the persistent sub-matrix kernel reads and maintains GT
    UpdateDesc udesc() const {
        UpdateDesc d; d.G = mG(); d.fields = fields; d.f_stride = (long)nt * n; d.expv = expv; d.invexpv = invexpv; d.v_stride = (long)nt * n;
        d.tabs = tabs; d.perm = rs_perm; d.kprop = rs_k; d.u = rs_u; d.rs_stride = (long)nt * n; d.Upanel = Upanel; d.Wpanel = Wpanel; d.Cpanel = Cpanel;
        d.panel_stride = (long)UPDATE_KD * n; d.state = state; d.state_stride = 4; d.prep = prep; d.prep_stride = 4L * n; d.slice_sync = (persistent && !handoff_failed) ? slice_sync : nullptr; d.slice_epoch = slice_epoch; d.slice_absent_tile = slice_absent_tile; d.slice_absent_l = slice_absent_l; d.slice_late_tile = slice_late_tile; d.slice_late_us = slice_late_us; d.GT = Mat{GT, nn}; d.gt_valid = gt_valid ? 1 : 0; d.info = ctx.info(); d.acc_out = acc; d.acc_stride = nt; d.n = n; d.nt = nt;
        return d;
    }
    int local_update(int l) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profiling) {
            if (ev_used == ev_pairs.size()) { hipEvent_t a, b; DQ_HIP(hipEventCreate(&a)); DQ_HIP(hipEventCreate(&b)); ev_pairs.emplace_back(a, b); }
            e0 = ev_pairs[ev_used].first; e1 = ev_pairs[ev_used].second; ++ev_used;
            DQ_HIP(hipEventRecord(e0, s));
        }
        if (persistent && !handoff_failed) {
            // every launch of a persistent slice kernel gets its own number: the hand-off words carry it, so none has to be re-armed
            if (++slice_epoch >= SLICE_EPOCH_LIMIT) { DQ_HIP(hipMemsetAsync(slice_sync, 0, (size_t)C * SLICE_SYNC_BYTES, s)); slice_epoch = 1; }
        }
        int gt_kept = 1;
        DQ_TRY(launch_update_slice(udesc(), l, l, C, s, &gt_kept));
        if (!gt_kept) gt_valid = false;                               // n > 256 on the kernel pairs: the next persistent launch transposes first
        if (profiling) DQ_HIP(hipEventRecord(e1, s));
        return 0;
    }
    int upload_stream(const int32_t* perm, const uint8_t* kprop, const double* u) {
        const size_t cnt = (size_t)C * nt * n;
        if (!perm || !kprop || !u) { set_error("sweep: null random-stream pointer"); return DQMC_EINVAL; }
        // the device indexes G with perm and the proposal table with kprop: every slice must carry a permutation of the sites and
        // proposal indices in {0, 1, 2} (what std::shuffle and uniform_int(0, 2) produce, source/update.cpp:14, include/field.h:79)
        seen.assign((size_t)n, 0);
        for (size_t row = 0; row < (size_t)C * nt; ++row) {
            const int32_t* pr = perm + row * n; const uint8_t* kr = kprop + row * n; const unsigned char mark = (unsigned char)(1 + (row & 1));
            if ((row & 1) == 0) std::fill(seen.begin(), seen.end(), 0);
            for (int idx = 0; idx < n; ++idx) {
                const int32_t i = pr[idx];
                if (i < 0 || i >= n || seen[i] == mark || kr[idx] > 2) { set_error("sweep: perm is not a permutation of the sites, or kprop > 2"); return DQMC_EINVAL; }
                seen[i] = mark;
            }
        }
        DQ_HIP(hipEventSynchronize(stage_free));                   // previous H2D copies out of the staging buffer are done
        char* base = static_cast<char*>(h_stage);
        double* hu = reinterpret_cast<double*>(base);
        int32_t* hp = reinterpret_cast<int32_t*>(base + cnt * sizeof(double));
        uint8_t* hk = reinterpret_cast<uint8_t*>(base + cnt * (sizeof(double) + sizeof(int32_t)));
        std::memcpy(hu, u, cnt * sizeof(double)); std::memcpy(hp, perm, cnt * sizeof(int32_t)); std::memcpy(hk, kprop, cnt);
        DQ_HIP(hipMemcpyAsync(rs_u, hu, cnt * sizeof(double), hipMemcpyHostToDevice, s));
        DQ_HIP(hipMemcpyAsync(rs_perm, hp, cnt * sizeof(int32_t), hipMemcpyHostToDevice, s));
        DQ_HIP(hipMemcpyAsync(rs_k, hk, cnt, hipMemcpyHostToDevice, s));
        DQ_HIP(hipEventRecord(stage_free, s));
        return 0;
    }
    // DQMC::sweep_0_to_beta (source/dqmc.cpp:337-396)
    int sweep_fwd() {
        int n_err = 0;
        DQ_HIP(hipMemsetAsync(err, 0, sizeof(double) * C * n_stack, s));             // max_abs_diff folds into zeroed slots
        const bool pg = piggyback();
        double* Pcur = bb0; double* Pnxt = bb1;
        for (int l = 0; l < nt; ++l) {
            const int is = stack_idx(l), loc = local_l(l);
            if (pg && loc >= 1) { DQ_TRY(wrap_forward_piggy(l, loc == 1 ? pg_eye : Pcur, Pnxt)); std::swap(Pcur, Pnxt); }   // P = B_{l-1} ... B_{l0}
            else DQ_TRY(wrap_forward(l));
            DQ_TRY(local_update(l));
            if (loc == loc_l_end[is]) {
                std::swap(G, Gtmp);                                                          // the wrapped G is kept for check_error, the stabilised one is written into the other buffer (no copy)
                Mat bb;
                if (pg) {                                                                     // the block's last factor, whose slice has only now been updated
                    if (loc == 0) DQ_TRY(launch_scale_rows(CMat(expK, nn), ev(l), Mat{Pnxt, nn}, n, C, s));
                    else DQ_TRY(ctx.gemm(CMat(expK, nn), CMat(Pcur, nn), Mat{Pnxt, nn}, ev(l)));
                    std::swap(Pcur, Pnxt); bb = Mat{Pcur, nn};
                } else DQ_TRY(Bbar(is, &bb));
                if (is == 0) DQ_TRY(ctx.to_ldr(bb, stk(0)));                                  // update_stack_forward :134-146
                else DQ_TRY(ctx.mat_mul_ldr(bb, stk(is - 1), stk(is)));
                gt_valid = false;                                                            // G is replaced below
                if (l == nt - 1) DQ_TRY(ctx.inv_I_plus_ldr(stk(is), mG(), logdet));           // stabilize_GF_forward :148-161
                else DQ_TRY(ctx.inv_I_plus_ldr_mul_ldr(stk(is), stk(is + 1), mG()));
                DQ_TRY(launch_max_abs_diff(CMat(Gtmp, nn), mG(), err + n_err, n_stack, n, C, s));   // check_error :317-329
                ++n_err;
            }
        }
        return launch_fold_stats(dstats, acc, nt, nt, err, n_stack, n_err, n, nt, C, s);
    }
    // DQMC::sweep_beta_to_0 (source/dqmc.cpp:398-456)
    int sweep_bwd() {
        int n_err = 0;
        DQ_HIP(hipMemsetAsync(err, 0, sizeof(double) * C * n_stack, s));
        const bool pg = piggyback();
        double* Pcur = bb0; double* Pnxt = bb1;
        for (int l = nt - 1; l >= 0; --l) {
            DQ_TRY(local_update(l));
            const int is = stack_idx(l);
            if (pg) { DQ_TRY(wrap_backward_piggy(l, local_l(l) == loc_l_end[is] ? pg_eye : Pcur, Pnxt)); std::swap(Pcur, Pnxt); }   // P = B_hi ... B_l
            else DQ_TRY(wrap_backward(l));
            if (local_l(l) == 0) {
                std::swap(G, Gtmp);
                Mat bb;
                if (pg) bb = Mat{Pcur, nn}; else DQ_TRY(Bbar(is, &bb));
                if (is == n_stack - 1) DQ_TRY(ctx.to_ldr(bb, stk(is)));                       // update_stack_backward :189-201
                else DQ_TRY(ctx.ldr_mul_mat(stk(is + 1), bb, stk(is)));
                gt_valid = false;
                if (l == 0) DQ_TRY(ctx.inv_I_plus_ldr(stk(is), mG(), logdet));                // stabilize_GF_backward :203-215
                else DQ_TRY(ctx.inv_I_plus_ldr_mul_ldr(stk(is - 1), stk(is), mG()));
                DQ_TRY(launch_max_abs_diff(CMat(Gtmp, nn), mG(), err + n_err, n_stack, n, C, s));
                ++n_err;
            }
        }
        return launch_fold_stats(dstats, acc, nt, nt, err, n_stack, n_err, n, nt, C, s);
    }
    // DQMC::sweep_unequalTime (source/dqmc.cpp:458-515) with propagate_unequalTime_GF_forward :223-248, propagate_Bt0_Bbt :250-264,
    // stabilize_unequalTime :266-285.  No Monte Carlo moves: B_l comes from the current fields (what the reference's B_ / invB_
    // caches hold after sweep_beta_to_0), Gtt[0] is the current G, Bbt = stack[i + 1] as the backward sweep left it.
    Mat utm(int which, int l) const { return Mat{utG[which] + (long)l * C * nn, nn}; }
    int sweep_unequal() {
        if (!utG[0]) {
            for (int w = 0; w < 3; ++w) DQ_TRY(dalloc(&utG[w], (size_t)(nt + 1) * C * nn));
            DQ_TRY(dalloc(&utTmp, (size_t)3 * C * nn)); DQ_TRY(dalloc(&utErr, (size_t)C * 3 * n_stack));
            const size_t mcount = (size_t)C * 3 * (nt + 1) * n;
            DQ_TRY(dalloc(&utMeasNow, mcount)); DQ_TRY(dalloc(&utMeasSum, mcount)); DQ_HIP(hipMemsetAsync(utMeasSum, 0, sizeof(double) * mcount, s));
            for (int b = 0; b < 2; ++b) { DQ_TRY(dalloc(&utL[b], C * nn)); DQ_TRY(dalloc(&utD[b], (size_t)C * n)); DQ_TRY(dalloc(&utR[b], C * nn)); DQ_TRY(dalloc(&utP[b], (size_t)C * n)); }
        }
        auto bt = [&](int b) { return LdrRef{Mat{utL[b], nn}, Vec{utD[b], (long)n}, Mat{utR[b], nn}, utP[b], &utTri[b]}; };
        int cur = 0, n_err = 0;
        DQ_HIP(hipMemsetAsync(utErr, 0, sizeof(double) * C * 3 * n_stack, s));
        DQ_TRY(launch_copy(mG(), utm(0, 0), nn, C, s));
        for (int l = 0; l < nt; ++l) {
            if (l == 0) {                                                                            // :234-239
                DQ_TRY(launch_copy(utm(0, 0), utm(1, 0), nn, C, s));
                DQ_TRY(launch_axpb_identity(utm(0, 0), utm(2, 0), 1.0, -1.0, n, C, s));
            }
            DQ_TRY(ctx.gemm(CMat(expK, nn), utm(0, l), ctx.T(0)));                                   // :240 Gtt = B Gtt B^-1
            DQ_TRY(ctx.gemm(ctx.T(0), CMat(invexpK, nn), utm(0, l + 1), ev(l), CVec(), iev(l)));
            DQ_TRY(ctx.gemm(CMat(expK, nn), utm(1, l), utm(1, l + 1), ev(l)));                       // :241 Gt0 = B Gt0
            DQ_TRY(ctx.gemm(utm(2, l), CMat(invexpK, nn), utm(2, l + 1), CVec(), CVec(), iev(l)));   // :242 G0t = G0t B^-1
            const int is = stack_idx(l);
            if (local_l(l) == loc_l_end[is]) {
                for (int w = 0; w < 3; ++w) DQ_TRY(launch_copy(utm(w, l + 1), Mat{utTmp + (long)w * C * nn, nn}, nn, C, s));
                Mat bb; DQ_TRY(Bbar(is, &bb));
                if (is == 0) DQ_TRY(ctx.to_ldr(bb, bt(cur)));                                        // :255-259
                else { DQ_TRY(ctx.mat_mul_ldr(bb, bt(cur), bt(cur ^ 1))); cur ^= 1; }
                if (l == nt - 1) {                                                                   // :267-276
                    DQ_TRY(ctx.inv_I_plus_ldr(bt(cur), utm(0, l + 1), logdet));
                    DQ_TRY(launch_axpb_identity(utm(0, l + 1), utm(1, l + 1), -1.0, 1.0, n, C, s));
                    DQ_TRY(launch_axpb_identity(utm(0, l + 1), utm(2, l + 1), -1.0, 0.0, n, C, s));
                } else {                                                                             // :278-282, Bbt = stack[is + 1]
                    DQ_TRY(ctx.inv_I_plus_ldr_mul_ldr(bt(cur), stk(is + 1), utm(0, l + 1)));
                    DQ_TRY(ctx.inv_invldr_plus_ldr(bt(cur), stk(is + 1), utm(1, l + 1), false));
                    DQ_TRY(ctx.inv_invldr_plus_ldr(stk(is + 1), bt(cur), utm(2, l + 1), true));
                }
                for (int w = 0; w < 3; ++w) {                                                        // check_error x 3, :502-507
                    DQ_TRY(launch_max_abs_diff(CMat(utTmp + (long)w * C * nn, nn), utm(w, l + 1), utErr + n_err, 3L * n_stack, n, C, s));
                    ++n_err;
                }
            }
        }
        ut_valid = true;
        return launch_fold_stats(dstats, acc, nt, 0, utErr, 3L * n_stack, n_err, n, nt, C, s);
    }
    int sync_and_check() {
        DQ_HIP(hipStreamSynchronize(s));
        if (profiling && ev_used) {
            for (size_t k = 0; k < ev_used; ++k) { float ms = 0.f; DQ_HIP(hipEventElapsedTime(&ms, ev_pairs[k].first, ev_pairs[k].second)); upd_ms += ms; }
            upd_launches += (long long)ev_used; ev_used = 0;
        }
        int h_info2[2] = {0, 0};
        DQ_HIP(hipMemcpy(h_info2, ctx.info(), 2 * sizeof(int), hipMemcpyDeviceToHost));
        const int h_info = h_info2[0];
        if (h_info & 4) { handoff_failed = true; gt_valid = false; set_error("persistent slice kernel: a workgroup that had checked in stopped answering (device fault or pre-emption beyond the spin bound); the chain state is undefined -- set the fields again and call dqmc_init; this engine uses the scan / flush kernel pairs from now on"); (void)hipMemset(ctx.info(), 0, sizeof(int)); (void)hipMemset(slice_sync, 0, (size_t)C * SLICE_SYNC_BYTES); slice_epoch = 0; return DQMC_ENUMERIC; }
        if (h_info & 8) {
            // n > 256: the census of slice_sm_kernel failed at slice info[1] - 1.  That launch and (device-side latch, update_sm.hip) every later
            // slice_sm_kernel launch enqueued behind it left their slices untouched; the wraps and stabilisations of the calls in flight ran.
            handoff_failed = true; gt_valid = false;
            char msg[640];
            snprintf(msg, sizeof msg, "persistent sub-matrix slice kernel: its workgroups did not all become resident at time slice %d (device shared with other "
                     "work?).  That slice and every later slice of the call(s) in flight proposed NOTHING (fields unchanged there, acceptance counts 0, their part of "
                     "the random stream is spent); wraps and stabilisations ran, so fields, G and the stack are consistent with each other, but the sweep is not the "
                     "one the stream describes from that slice on.  This engine uses the scan / flush kernel pairs from now on: restore the fields of the last "
                     "completed sweep and call dqmc_init, or keep the state as a valid (shortened) sweep", h_info2[1] - 1);
            set_error(msg); (void)hipMemset(ctx.info(), 0, 2 * sizeof(int)); return DQMC_ENUMERIC;
        }
        if (h_info & 2) { set_error("cooperative QRCP gave up waiting for a partner workgroup (not co-resident?)"); (void)hipMemset(ctx.info(), 0, sizeof(int)); return DQMC_ENUMERIC; }
        if (h_info) { set_error("LU factorisation hit a zero or NaN pivot"); (void)hipMemset(ctx.info(), 0, sizeof(int)); return DQMC_ENUMERIC; }
        return 0;
    }
};

// ---- stateless calls: one cached single-chain context per n --------------------------------
static std::mutex g_ctx_mu;
static std::map<int, std::unique_ptr<Ctx>> g_ctx;
static std::map<int, double*> g_ctx_extra;     // 3 LDR triples worth of scratch per n

static int get_ctx(int n, Ctx** out, double** extra) {
    if (n <= 0) { set_error("n must be positive"); return DQMC_EINVAL; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { set_error("no HIP device available: this library requires a gfx950 GPU"); return DQMC_ENODEVICE; }
    auto it = g_ctx.find(n);
    if (it == g_ctx.end()) {
        std::unique_ptr<Ctx> c(new Ctx);
        DQ_TRY(c->init(n, 1, 0));
        double* ex = nullptr;
        DQ_HIP(hipMalloc(&ex, sizeof(double) * (8L * n * n + 4L * n)));
        g_ctx_extra[n] = ex;
        it = g_ctx.emplace(n, std::move(c)).first;
    }
    *out = it->second.get(); *extra = g_ctx_extra[n];
    return hipSetDevice(0) == hipSuccess ? 0 : DQMC_ENODEVICE;
}
struct HostLdr { const double* L; const double* d; const double* R; };
static int upload_ldr(Ctx* c, double* base, HostLdr h, LdrRef* out) {
    const long nn = c->nn; const int n = c->n;
    out->L = Mat{base, nn}; out->R = Mat{base + nn, nn}; out->d = Vec{base + 2 * nn, (long)n};
    DQ_HIP(hipMemcpyAsync(out->L.p, h.L, sizeof(double) * nn, hipMemcpyHostToDevice, c->stream));
    DQ_HIP(hipMemcpyAsync(out->R.p, h.R, sizeof(double) * nn, hipMemcpyHostToDevice, c->stream));
    DQ_HIP(hipMemcpyAsync(out->d.p, h.d, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    return 0;
}
static int download_ldr(Ctx* c, LdrRef f, double* L, double* d, double* R) {
    DQ_HIP(hipMemcpyAsync(L, f.L.p, sizeof(double) * c->nn, hipMemcpyDeviceToHost, c->stream));
    DQ_HIP(hipMemcpyAsync(R, f.R.p, sizeof(double) * c->nn, hipMemcpyDeviceToHost, c->stream));
    DQ_HIP(hipMemcpyAsync(d, f.d.p, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
    DQ_HIP(hipStreamSynchronize(c->stream));
    int h_info = 0;
    DQ_HIP(hipMemcpy(&h_info, c->info(), sizeof(int), hipMemcpyDeviceToHost));
    if (h_info) { set_error("factorisation hit a zero or NaN pivot"); (void)hipMemset(c->info(), 0, sizeof(int)); return DQMC_ENUMERIC; }
    return 0;
}
static LdrRef slot(Ctx* c, double* extra, int k) {     // k = 0,1,2: LDR-sized scratch slots after the two upload slots
    double* base = extra + (long)k * (2 * c->nn + c->n);
    return LdrRef{Mat{base, c->nn}, Vec{base + 2 * c->nn, (long)c->n}, Mat{base + c->nn, c->nn}};
}

}  // namespace dq

using namespace dq;

struct dqmc_engine { Engine e; };

namespace dq {
int engine_fields_view(dqmc_engine* h, EngineFieldsView* v) {
    if (!h || !v) { set_error("null engine"); return DQMC_EINVAL; }
    Engine& e = h->e;
    v->device = e.device; v->n = e.n; v->nt = e.nt; v->n_chains = e.C; v->fields = e.fields; v->stream = e.s;
    return 0;
}
int engine_fields_changed(dqmc_engine* h) {
    if (!h) { set_error("null engine"); return DQMC_EINVAL; }
    Engine& e = h->e;
    e.stack_valid = false; e.gt_valid = false;
    return launch_build_expv(e.fields, (long)e.nt * e.n, e.nt, e.n, e.tab8, e.expv, e.invexpv, (long)e.nt * e.n, e.C, e.s);
}
}  // namespace dq

#define API_LOCK std::lock_guard<std::mutex> _lk(g_ctx_mu)
#define CHECK_E(e) if (!(e)) { set_error("null engine"); return DQMC_EINVAL; }

extern "C" {

const char* dqmc_last_error(void) { return get_error(); }
const char* dqmc_backend(void) { return "hip:gfx950"; }
int dqmc_device_count(void) { int c = 0; if (hipGetDeviceCount(&c) != hipSuccess) return 0; return c; }

int dqmc_to_ldr(int n, const double* M, double* L, double* d, double* R) {
    API_LOCK; Ctx* c; double* ex; DQ_TRY(get_ctx(n, &c, &ex));
    if (!M || !L || !d || !R) { set_error("null pointer"); return DQMC_EINVAL; }
    DQ_HIP(hipMemcpyAsync(c->T(5).p, M, sizeof(double) * c->nn, hipMemcpyHostToDevice, c->stream));
    LdrRef out = slot(c, ex, 0);
    DQ_TRY(c->to_ldr(c->T(5), out));
    return download_ldr(c, out, L, d, R);
}
int dqmc_ldr_mul_mat(int n, const double* L, const double* d, const double* R, const double* M, double* Lo, double* d_o, double* Ro) {
    API_LOCK; Ctx* c; double* ex; DQ_TRY(get_ctx(n, &c, &ex));
    LdrRef F; DQ_TRY(upload_ldr(c, slot(c, ex, 0).L.p, HostLdr{L, d, R}, &F));
    DQ_HIP(hipMemcpyAsync(c->T(5).p, M, sizeof(double) * c->nn, hipMemcpyHostToDevice, c->stream));
    LdrRef out = slot(c, ex, 2);
    DQ_TRY(c->ldr_mul_mat(F, c->T(5), out));
    return download_ldr(c, out, Lo, d_o, Ro);
}
int dqmc_mat_mul_ldr(int n, const double* M, const double* L, const double* d, const double* R, double* Lo, double* d_o, double* Ro) {
    API_LOCK; Ctx* c; double* ex; DQ_TRY(get_ctx(n, &c, &ex));
    LdrRef F; DQ_TRY(upload_ldr(c, slot(c, ex, 0).L.p, HostLdr{L, d, R}, &F));
    DQ_HIP(hipMemcpyAsync(c->T(5).p, M, sizeof(double) * c->nn, hipMemcpyHostToDevice, c->stream));
    LdrRef out = slot(c, ex, 2);
    DQ_TRY(c->mat_mul_ldr(c->T(5), F, out));
    return download_ldr(c, out, Lo, d_o, Ro);
}
int dqmc_ldr_mul_ldr(int n, const double* L1, const double* d1, const double* R1, const double* L2, const double* d2, const double* R2, double* Lo, double* d_o, double* Ro) {
    API_LOCK; Ctx* c; double* ex; DQ_TRY(get_ctx(n, &c, &ex));
    LdrRef F1, F2; DQ_TRY(upload_ldr(c, slot(c, ex, 0).L.p, HostLdr{L1, d1, R1}, &F1)); DQ_TRY(upload_ldr(c, slot(c, ex, 1).L.p, HostLdr{L2, d2, R2}, &F2));
    LdrRef out = slot(c, ex, 2);
    DQ_TRY(c->ldr_mul_ldr(F1, F2, out));
    return download_ldr(c, out, Lo, d_o, Ro);
}
int dqmc_inv_I_plus_ldr(int n, const double* L, const double* d, const double* R, double* G, double* logdet) {
    API_LOCK; Ctx* c; double* ex; DQ_TRY(get_ctx(n, &c, &ex));
    LdrRef F; DQ_TRY(upload_ldr(c, slot(c, ex, 0).L.p, HostLdr{L, d, R}, &F));
    DQ_TRY(c->inv_I_plus_ldr(F, c->T(6), c->scal(1)));
    DQ_HIP(hipMemcpyAsync(G, c->T(6).p, sizeof(double) * c->nn, hipMemcpyDeviceToHost, c->stream));
    double ld = 0.0;
    DQ_HIP(hipMemcpyAsync(&ld, c->scal(1), sizeof(double), hipMemcpyDeviceToHost, c->stream));
    DQ_HIP(hipStreamSynchronize(c->stream));
    if (logdet) *logdet = ld;
    int h_info = 0; DQ_HIP(hipMemcpy(&h_info, c->info(), sizeof(int), hipMemcpyDeviceToHost));
    if (h_info) { set_error("LU hit a zero or NaN pivot"); (void)hipMemset(c->info(), 0, sizeof(int)); return DQMC_ENUMERIC; }
    return 0;
}
int dqmc_inv_I_plus_ldr_mul_ldr(int n, const double* L1, const double* d1, const double* R1, const double* L2, const double* d2, const double* R2, double* G) {
    API_LOCK; Ctx* c; double* ex; DQ_TRY(get_ctx(n, &c, &ex));
    LdrRef F1, F2; DQ_TRY(upload_ldr(c, slot(c, ex, 0).L.p, HostLdr{L1, d1, R1}, &F1)); DQ_TRY(upload_ldr(c, slot(c, ex, 1).L.p, HostLdr{L2, d2, R2}, &F2));
    DQ_TRY(c->inv_I_plus_ldr_mul_ldr(F1, F2, c->T(6)));
    DQ_HIP(hipMemcpyAsync(G, c->T(6).p, sizeof(double) * c->nn, hipMemcpyDeviceToHost, c->stream));
    DQ_HIP(hipStreamSynchronize(c->stream));
    int h_info = 0; DQ_HIP(hipMemcpy(&h_info, c->info(), sizeof(int), hipMemcpyDeviceToHost));
    if (h_info) { set_error("LU hit a zero or NaN pivot"); (void)hipMemset(c->info(), 0, sizeof(int)); return DQMC_ENUMERIC; }
    return 0;
}
int dqmc_gemm(int n, const double* A, int transA, const double* B, int transB, double* Cm) {
    API_LOCK; Ctx* c; double* ex; DQ_TRY(get_ctx(n, &c, &ex));
    DQ_HIP(hipMemcpyAsync(c->T(5).p, A, sizeof(double) * c->nn, hipMemcpyHostToDevice, c->stream));
    DQ_HIP(hipMemcpyAsync(c->T(6).p, B, sizeof(double) * c->nn, hipMemcpyHostToDevice, c->stream));
    CMat Bm = c->T(6);
    if (transB) { DQ_TRY(launch_transpose_scale(c->T(6), c->T(7), CVec(), n, 1, c->stream)); Bm = c->T(7); }
    DQ_TRY(c->gemm(c->T(5), Bm, c->T(8), CVec(), CVec(), CVec(), transA ? 1 : 0));
    DQ_HIP(hipMemcpyAsync(Cm, c->T(8).p, sizeof(double) * c->nn, hipMemcpyDeviceToHost, c->stream));
    DQ_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int dqmc_rank1_update(int n, double* G, int i, double delta) {
    API_LOCK; Ctx* c; double* ex; DQ_TRY(get_ctx(n, &c, &ex));
    if (i < 0 || i >= n) { set_error("site index out of range"); return DQMC_ERANGE; }
    DQ_HIP(hipMemcpyAsync(c->T(5).p, G, sizeof(double) * c->nn, hipMemcpyHostToDevice, c->stream));
    DQ_TRY(launch_rank1(c->T(5), i, delta, c->T(6).p, 2L * n + 1, n, 1, c->stream));
    DQ_HIP(hipMemcpyAsync(G, c->T(5).p, sizeof(double) * c->nn, hipMemcpyDeviceToHost, c->stream));
    DQ_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int dqmc_create_batch(dqmc_engine** out, int device, int n_chains, int n_sites, int nt, int n_stab, const double* g,
                      const double gamma[4], const double eta[4], const double* expK, const double* invexpK) {
    if (!out || n_chains <= 0 || n_sites <= 0 || nt <= 0 || n_stab <= 0 || !g || !gamma || !eta || !expK || !invexpK) { set_error("bad argument"); return DQMC_EINVAL; }
    if (n_sites > 1024) { set_error("n_sites > 1024 is not supported by the single-workgroup kernels"); return DQMC_EINVAL; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { set_error("no HIP device available: this library requires a gfx950 GPU"); return DQMC_ENODEVICE; }
    if (device < 0 || device >= count) { set_error("device ordinal out of range"); return DQMC_EINVAL; }
    dqmc_engine* h = new (std::nothrow) dqmc_engine;
    if (!h) return DQMC_ENOMEM;
    int rc = h->e.create(device, n_chains, n_sites, nt, n_stab, g, gamma, eta, expK, invexpK);
    if (rc) { delete h; return rc; }
    *out = h; return 0;
}
int dqmc_create(dqmc_engine** out, int device, int n_sites, int nt, int n_stab, double g, const double gamma[4], const double eta[4],
                const double* expK, const double* invexpK) {
    return dqmc_create_batch(out, device, 1, n_sites, nt, n_stab, &g, gamma, eta, expK, invexpK);
}
int dqmc_set_checkerboard(dqmc_engine* h, int n_groups, const int32_t* bonds, const int32_t* group_sizes, const double* cosh_t, const double* sinh_t,
                          const double* diag_factor) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    return e.set_checkerboard(n_groups, bonds, group_sizes, cosh_t, sinh_t, diag_factor);
}
void dqmc_destroy(dqmc_engine* h) { if (h) { (void)hipSetDevice(h->e.device); delete h; } }
int dqmc_n_chains(dqmc_engine* h) { return h ? h->e.C : 0; }

int dqmc_set_fields(dqmc_engine* h, const int64_t* f) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    std::vector<int8_t> tmp((size_t)e.C * e.nt * e.n);
    for (int c = 0; c < e.C; ++c) for (int i = 0; i < e.n; ++i) for (int l = 0; l < e.nt; ++l) {
        const int64_t v = f[(size_t)c * e.nt * e.n + l + (size_t)e.nt * i];
        if (v < 0 || v > 3) { set_error("field value outside {0,1,2,3}"); return DQMC_EINVAL; }
        tmp[(size_t)c * e.nt * e.n + (size_t)l * e.n + i] = (int8_t)v;
    }
    DQ_HIP(hipStreamSynchronize(e.s));
    DQ_HIP(hipMemcpy(e.fields, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    DQ_TRY(launch_build_expv(e.fields, (long)e.nt * e.n, e.nt, e.n, e.tab8, e.expv, e.invexpv, (long)e.nt * e.n, e.C, e.s));
    DQ_HIP(hipStreamSynchronize(e.s));
    return 0;
}
int dqmc_get_fields(dqmc_engine* h, int64_t* f) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    DQ_TRY(e.sync_and_check());
    std::vector<int8_t> tmp((size_t)e.C * e.nt * e.n);
    DQ_HIP(hipMemcpy(tmp.data(), e.fields, tmp.size(), hipMemcpyDeviceToHost));
    for (int c = 0; c < e.C; ++c) for (int i = 0; i < e.n; ++i) for (int l = 0; l < e.nt; ++l)
        f[(size_t)c * e.nt * e.n + l + (size_t)e.nt * i] = tmp[(size_t)c * e.nt * e.n + (size_t)l * e.n + i];
    return 0;
}
int dqmc_init(dqmc_engine* h) { CHECK_E(h); DQ_HIP(hipSetDevice(h->e.device)); DQ_TRY(h->e.init()); return h->e.sync_and_check(); }
int dqmc_get_G(dqmc_engine* h, double* G) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device)); DQ_TRY(e.sync_and_check());
    DQ_HIP(hipMemcpy(G, e.G, sizeof(double) * e.C * e.nn, hipMemcpyDeviceToHost)); return 0;
}
int dqmc_set_G(dqmc_engine* h, const double* G) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device)); DQ_HIP(hipStreamSynchronize(e.s));
    DQ_HIP(hipMemcpy(e.G, G, sizeof(double) * e.C * e.nn, hipMemcpyHostToDevice)); e.gt_valid = false; return 0;
}
int dqmc_get_logdet(dqmc_engine* h, double* ld) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device)); DQ_TRY(e.sync_and_check());
    DQ_HIP(hipMemcpy(ld, e.logdet, sizeof(double) * e.C, hipMemcpyDeviceToHost)); return 0;
}
int dqmc_n_stack(dqmc_engine* h) { return h ? h->e.n_stack : 0; }
int dqmc_get_stack(dqmc_engine* h, int i, double* L, double* d, double* R) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (i < 0 || i >= e.n_stack) { set_error("LDR Stack index out of bounds"); return DQMC_ERANGE; }
    if (!e.stack_valid) { set_error("stack not initialised: call dqmc_init first"); return DQMC_EINVAL; }
    DQ_TRY(e.sync_and_check());
    LdrRef f = e.stk(i);
    DQ_HIP(hipMemcpy(L, f.L.p, sizeof(double) * e.C * e.nn, hipMemcpyDeviceToHost));
    DQ_HIP(hipMemcpy(R, f.R.p, sizeof(double) * e.C * e.nn, hipMemcpyDeviceToHost));
    DQ_HIP(hipMemcpy(d, f.d.p, sizeof(double) * e.C * e.n, hipMemcpyDeviceToHost));
    return 0;
}
int dqmc_sweep_0_to_beta(dqmc_engine* h, const int32_t* perm, const uint8_t* kprop, const double* u) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (!e.stack_valid) { set_error("dqmc_init must be called before sweeping"); return DQMC_EINVAL; }
    DQ_TRY(e.upload_stream(perm, kprop, u)); return e.sweep_fwd();
}
int dqmc_sweep_beta_to_0(dqmc_engine* h, const int32_t* perm, const uint8_t* kprop, const double* u) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (!e.stack_valid) { set_error("dqmc_init must be called before sweeping"); return DQMC_EINVAL; }
    DQ_TRY(e.upload_stream(perm, kprop, u)); return e.sweep_bwd();
}
int dqmc_sync(dqmc_engine* h) { CHECK_E(h); DQ_HIP(hipSetDevice(h->e.device)); return h->e.sync_and_check(); }
int dqmc_get_stats(dqmc_engine* h, dqmc_stats* out) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device)); DQ_TRY(e.sync_and_check());
    static_assert(sizeof(DevStats) == sizeof(dqmc_stats), "stats layout");
    DQ_HIP(hipMemcpy(out, e.dstats, sizeof(DevStats) * e.C, hipMemcpyDeviceToHost)); return 0;
}
int dqmc_wrap_forward(dqmc_engine* h, int l) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (l < 0 || l >= e.nt) { set_error("time slice out of range"); return DQMC_ERANGE; }
    return e.wrap_forward(l);
}
int dqmc_wrap_backward(dqmc_engine* h, int l) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (l < 0 || l >= e.nt) { set_error("time slice out of range"); return DQMC_ERANGE; }
    return e.wrap_backward(l);
}
int dqmc_local_update_slice(dqmc_engine* h, int l, const int32_t* perm, const uint8_t* kprop, const double* u, int* accepted) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (l < 0 || l >= e.nt) { set_error("time slice out of range"); return DQMC_ERANGE; }
    if (!perm || !kprop || !u) { set_error("local_update_slice: null random-stream pointer"); return DQMC_EINVAL; }
    for (int c = 0; c < e.C; ++c) {                                          // same contract as the sweeps (Engine::upload_stream)
        std::vector<unsigned char> seen((size_t)e.n, 0);
        for (int idx = 0; idx < e.n; ++idx) {
            const int32_t i = perm[(size_t)c * e.n + idx];
            if (i < 0 || i >= e.n || seen[i] || kprop[(size_t)c * e.n + idx] > 2) { set_error("local_update_slice: perm is not a permutation of the sites, or kprop > 2"); return DQMC_EINVAL; }
            seen[i] = 1;
        }
    }
    DQ_HIP(hipStreamSynchronize(e.s));
    for (int c = 0; c < e.C; ++c) {
        const size_t off = ((size_t)c * e.nt + l) * e.n;
        DQ_HIP(hipMemcpy(e.rs_perm + off, perm + (size_t)c * e.n, sizeof(int32_t) * e.n, hipMemcpyHostToDevice));
        DQ_HIP(hipMemcpy(e.rs_k + off, kprop + (size_t)c * e.n, e.n, hipMemcpyHostToDevice));
        DQ_HIP(hipMemcpy(e.rs_u + off, u + (size_t)c * e.n, sizeof(double) * e.n, hipMemcpyHostToDevice));
    }
    DQ_TRY(e.local_update(l));
    DQ_TRY(launch_fold_stats(e.dstats, e.acc + l, e.nt, 1, e.err, e.n_stack, 0, e.n, e.nt, e.C, e.s));
    DQ_TRY(e.sync_and_check());
    if (accepted) for (int c = 0; c < e.C; ++c) DQ_HIP(hipMemcpy(accepted + c, e.acc + (size_t)c * e.nt + l, sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}
int dqmc_calculate_Bbar(dqmc_engine* h, int is, double* out) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (is < 0 || is >= e.n_stack) { set_error("stack index out of range"); return DQMC_ERANGE; }
    Mat bb; DQ_TRY(e.Bbar(is, &bb)); DQ_HIP(hipStreamSynchronize(e.s));
    DQ_HIP(hipMemcpy(out, bb.p, sizeof(double) * e.C * e.nn, hipMemcpyDeviceToHost)); return 0;
}
// AttractiveHubbard::global_action (source/model.cpp:140-159); the field sums run on the host copy
int dqmc_global_action(dqmc_engine* h, double* S) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device)); DQ_TRY(e.sync_and_check());
    std::vector<int8_t> tmp((size_t)e.C * e.nt * e.n); std::vector<double> ld(e.C);
    DQ_HIP(hipMemcpy(tmp.data(), e.fields, tmp.size(), hipMemcpyDeviceToHost));
    DQ_HIP(hipMemcpy(ld.data(), e.logdet, sizeof(double) * e.C, hipMemcpyDeviceToHost));
    const double alpha = -1.0;
    for (int c = 0; c < e.C; ++c) {
        double lb = 0.0, lg = 0.0;
        for (int i = 0; i < e.n; ++i) for (int l = 0; l < e.nt; ++l) {     // arma::imat memory order
            const int f = tmp[(size_t)c * e.nt * e.n + (size_t)l * e.n + i];
            lb += alpha * e.g_host[c] * e.eta_host[f]; lg += std::log(e.gamma_host[f]);
        }
        S[c] = -2.0 * ld[c] - (lb + lg);
    }
    return 0;
}
int dqmc_sweep_unequal_time(dqmc_engine* h) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (!e.stack_valid) { set_error("sweep_unequal_time: call dqmc_init first"); return DQMC_EINVAL; }
    return e.sweep_unequal();
}
int dqmc_get_G_tau(dqmc_engine* h, int which, int l, double* out) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (which < 0 || which > 2 || l < 0 || l > e.nt) { set_error("get_G_tau: which in 0..2, l in 0..nt"); return DQMC_ERANGE; }
    if (!e.ut_valid) { set_error("get_G_tau: run dqmc_sweep_unequal_time first"); return DQMC_EINVAL; }
    DQ_TRY(e.sync_and_check());
    DQ_HIP(hipMemcpy(out, e.utG[which] + (long)l * e.C * e.nn, sizeof(double) * e.C * e.nn, hipMemcpyDeviceToHost));
    return 0;
}
// DQMC::half_warp (source/dqmc.cpp:288-315): invexpK_half * M * expK_half, two GEMMs on the engine's stream
int dqmc_half_warp(dqmc_engine* h, const double* expK_half, const double* invexpK_half, int which, int l, double* out) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (!out) { set_error("half_warp: out is NULL"); return DQMC_EINVAL; }
    if (which < -1 || which > 2 || (which >= 0 && (l < 0 || l > e.nt))) { set_error("half_warp: which in -1..2, l in 0..nt"); return DQMC_ERANGE; }
    if (which >= 0 && !e.ut_valid) { set_error("half_warp: run dqmc_sweep_unequal_time first"); return DQMC_EINVAL; }
    if ((expK_half == nullptr) != (invexpK_half == nullptr)) { set_error("half_warp: pass both half-step matrices or neither"); return DQMC_EINVAL; }
    if (expK_half) {
        if (!e.expKh) { DQ_TRY(e.dalloc(&e.expKh, e.nn)); DQ_TRY(e.dalloc(&e.invexpKh, e.nn)); DQ_TRY(e.dalloc(&e.hwOut, (size_t)e.C * e.nn)); }
        DQ_HIP(hipStreamSynchronize(e.s));                         // pageable host memory: the copies below are synchronous with respect to the host
        DQ_HIP(hipMemcpy(e.expKh, expK_half, sizeof(double) * e.nn, hipMemcpyHostToDevice));
        DQ_HIP(hipMemcpy(e.invexpKh, invexpK_half, sizeof(double) * e.nn, hipMemcpyHostToDevice));
    } else if (!e.expKh) { set_error("half_warp: no half-step matrices uploaded yet"); return DQMC_EINVAL; }
    const CMat M = which < 0 ? CMat(e.G, e.nn) : CMat(e.utG[which] + (long)l * e.C * e.nn, e.nn);
    DQ_TRY(e.ctx.gemm(CMat(e.invexpKh, 0), M, e.ctx.T(0)));                     // chain stride 0: one matrix for every chain
    DQ_TRY(e.ctx.gemm(e.ctx.T(0), CMat(e.expKh, 0), Mat{e.hwOut, e.nn}));
    DQ_TRY(e.sync_and_check());
    DQ_HIP(hipMemcpy(out, e.hwOut, sizeof(double) * e.C * e.nn, hipMemcpyDeviceToHost));
    return 0;
}
int dqmc_measure_unequal_time(dqmc_engine* h, int L1, int L2, int accumulate, double* out) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (L1 < 1 || L2 < 1 || L1 * L2 != e.n) { set_error("measure_unequal_time: L1*L2 must equal n_sites"); return DQMC_EINVAL; }
    if (!e.ut_valid) { set_error("measure_unequal_time: run dqmc_sweep_unequal_time first"); return DQMC_EINVAL; }
    const long stride = 3L * (e.nt + 1) * e.n;
    if (accumulate) {
        DQ_TRY(launch_measure_unequal_time(e.utG[0], e.utG[1], e.utG[2], e.utMeasSum, stride, L1, L2, e.nt, 1, e.C, e.s));
        ++e.ut_meas_count; return 0;
    }
    if (!out) { set_error("measure_unequal_time: out is NULL"); return DQMC_EINVAL; }
    DQ_TRY(launch_measure_unequal_time(e.utG[0], e.utG[1], e.utG[2], e.utMeasNow, stride, L1, L2, e.nt, 0, e.C, e.s));
    DQ_TRY(e.sync_and_check());
    DQ_HIP(hipMemcpy(out, e.utMeasNow, sizeof(double) * e.C * stride, hipMemcpyDeviceToHost));
    return 0;
}
int dqmc_measure_unequal_fetch(dqmc_engine* h, double* out_sum, int64_t* n_measurements, int reset) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device)); DQ_TRY(e.sync_and_check());
    if (!e.utMeasSum) { set_error("measure_unequal_fetch: nothing measured yet"); return DQMC_EINVAL; }
    const size_t cnt = (size_t)e.C * 3 * (e.nt + 1) * e.n;
    if (out_sum) DQ_HIP(hipMemcpy(out_sum, e.utMeasSum, sizeof(double) * cnt, hipMemcpyDeviceToHost));
    if (n_measurements) *n_measurements = e.ut_meas_count;
    if (reset) { DQ_HIP(hipMemset(e.utMeasSum, 0, sizeof(double) * cnt)); e.ut_meas_count = 0; }
    return 0;
}
int dqmc_measure_equal_time(dqmc_engine* h, int L1, int L2, double* scalars, double* chi_r) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (L1 < 1 || L2 < 1 || L1 * L2 != e.n) { set_error("measure_equal_time: L1*L2 must equal n_sites"); return DQMC_EINVAL; }
    DQ_TRY(launch_measure_equal_time(CMat(e.G, e.nn), e.meas_now, 3 + e.n, L1, L2, 0, e.C, e.s));
    DQ_TRY(e.sync_and_check());
    std::vector<double> tmp((size_t)e.C * (3 + e.n));
    DQ_HIP(hipMemcpy(tmp.data(), e.meas_now, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
    for (int c = 0; c < e.C; ++c) {
        if (scalars) std::copy(tmp.begin() + (size_t)c * (3 + e.n), tmp.begin() + (size_t)c * (3 + e.n) + 3, scalars + 3 * (size_t)c);
        if (chi_r) std::copy(tmp.begin() + (size_t)c * (3 + e.n) + 3, tmp.begin() + (size_t)(c + 1) * (3 + e.n), chi_r + (size_t)c * e.n);
    }
    return 0;
}
int dqmc_measure_accumulate(dqmc_engine* h, int L1, int L2) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    if (L1 < 1 || L2 < 1 || L1 * L2 != e.n) { set_error("measure_accumulate: L1*L2 must equal n_sites"); return DQMC_EINVAL; }
    DQ_TRY(launch_measure_equal_time(CMat(e.G, e.nn), e.meas_sum, 3 + e.n, L1, L2, 1, e.C, e.s));     // asynchronous, in stream order after the sweep
    ++e.meas_count;
    return 0;
}
int dqmc_measure_fetch(dqmc_engine* h, double* scalars_sum, double* chi_r_sum, int64_t* n_measurements, int reset) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device)); DQ_TRY(e.sync_and_check());
    std::vector<double> tmp((size_t)e.C * (3 + e.n));
    DQ_HIP(hipMemcpy(tmp.data(), e.meas_sum, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
    for (int c = 0; c < e.C; ++c) {
        if (scalars_sum) std::copy(tmp.begin() + (size_t)c * (3 + e.n), tmp.begin() + (size_t)c * (3 + e.n) + 3, scalars_sum + 3 * (size_t)c);
        if (chi_r_sum) std::copy(tmp.begin() + (size_t)c * (3 + e.n) + 3, tmp.begin() + (size_t)(c + 1) * (3 + e.n), chi_r_sum + (size_t)c * e.n);
    }
    if (n_measurements) *n_measurements = e.meas_count;
    if (reset) { DQ_HIP(hipMemset(e.meas_sum, 0, sizeof(double) * tmp.size())); e.meas_count = 0; }
    return 0;
}
int dqmc_update_kernel_time(dqmc_engine* h, double* ms, int64_t* n_launches, int64_t* n_accepted) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device)); DQ_TRY(e.sync_and_check());
    std::vector<DevStats> st(e.C);
    DQ_HIP(hipMemcpy(st.data(), e.dstats, sizeof(DevStats) * e.C, hipMemcpyDeviceToHost));
    long long acc = 0; for (auto& x : st) acc += x.n_accepted;
    if (ms) *ms = e.upd_ms; if (n_launches) *n_launches = e.upd_launches; if (n_accepted) *n_accepted = acc - e.upd_accept_base;
    e.upd_ms = 0.0; e.upd_launches = 0; e.upd_accept_base = acc;
    return 0;
}
// diagnostic: 1 when the next local update of this engine takes a persistent single-launch slice kernel (it holds a CU reservation
// and no hand-off has failed), 0 for the kernel pairs
int dqmc_debug_snapshot(dqmc_engine* h, double* wrap_err, int* accepted, unsigned int* sync_words, unsigned int* slice_epoch) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device));
    DQ_HIP(hipStreamSynchronize(e.s));
    if (wrap_err) DQ_HIP(hipMemcpy(wrap_err, e.err, sizeof(double) * e.n_stack, hipMemcpyDeviceToHost));
    if (accepted) DQ_HIP(hipMemcpy(accepted, e.acc, sizeof(int) * e.nt, hipMemcpyDeviceToHost));
    if (sync_words) DQ_HIP(hipMemcpy(sync_words, e.slice_sync, sizeof(unsigned int) * 80, hipMemcpyDeviceToHost));
    if (slice_epoch) *slice_epoch = e.slice_epoch;
    return 0;
}
int dqmc_slice_path(dqmc_engine* h) {
    if (!h) return -1;
    Engine& e = h->e;
    if (!(e.persistent && !e.handoff_failed)) return 0;
    // 2: at least one launch of the persistent kernel fell back to the solo walk (a flush workgroup had not become resident in time)
    unsigned solo = 0;
    for (int c = 0; c < e.C && !solo; ++c) {
        SliceSync hs; if (hipMemcpy(&hs, e.slice_sync + (size_t)c * SLICE_SYNC_BYTES, 64, hipMemcpyDeviceToHost) != hipSuccess) break;
        solo = hs.solo_count;
    }
    return solo ? 2 : 1;
}
int dqmc_set_profiling(dqmc_engine* h, int on) {
    CHECK_E(h); Engine& e = h->e; DQ_HIP(hipSetDevice(e.device)); DQ_TRY(e.sync_and_check());
    e.profiling = on != 0; return 0;
}

}  // extern "C"
