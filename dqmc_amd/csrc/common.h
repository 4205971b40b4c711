// common.h -- shared declarations of the gfx950 DQMC sweep library.
//
// Everything here is device-side plumbing for the kernels in this directory:
// batched launch descriptors (every kernel takes a leading "chain" dimension
// in blockIdx.y so one launch advances all Markov chains an engine owns),
// error handling, and the launcher prototypes the engine (engine.hip) calls.
// gfx950 only: wave = 64 lanes, fp64 MFMA 16x16x4.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

struct dqmc_engine;

namespace dq {

constexpr int WAVE = 64;

// ---- error handling -------------------------------------------------------
void set_error(const std::string& msg);          // thread-local last error
const char* get_error();

#define DQ_HIP(call)                                                                    \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            ::dq::set_error(std::string(#call) + ": " + hipGetErrorString(_e));         \
            return -2; /* DQMC_ENODEVICE */                                             \
        }                                                                               \
    } while (0)

#define DQ_TRY_RC(expr)             \
    do {                            \
        int _rc = (expr);           \
        if (_rc != 0) return _rc;   \
    } while (0)

// A batched column-major matrix: chain c lives at p + c*stride (stride 0 =
// one matrix shared by all chains, e.g. expK when every chain has the same
// beta).  Leading dimension is always n.
struct Mat {
    double* p = nullptr;
    long stride = 0;
    __host__ __device__ double* at(int c) const { return p + (long)c * stride; }
};
struct CMat {
    const double* p = nullptr;
    long stride = 0;
    CMat() = default;
    __host__ __device__ CMat(const double* p_, long s) : p(p_), stride(s) {}
    __host__ __device__ CMat(const Mat& m) : p(m.p), stride(m.stride) {}
    __host__ __device__ const double* at(int c) const { return p + (long)c * stride; }
};
using Vec = Mat;     // batched n-vectors use the same (pointer, stride) pair
using CVec = CMat;

// ---- gemm.hip ---------------------------------------------------------------
// C[c] = diag(rs) * ( op(A[c]) * diag(ks) * B[c] ) * diag(cs)  (+ C[c] if accumulate)
// rs/ks/cs may be null (= identity).  op(A) = A or A^T.  n x n x n, fp64 MFMA.
struct GemmDesc {
    CMat A, B;
    Mat C;
    Mat CT{nullptr, 0};   // optional: also receives C^T (fused in the split-K epilogue, a transpose launch otherwise)
    CVec rs, ks, cs;
    int n = 0;
    int transA = 0;
    int accumulate = 0;
};
int launch_gemm(const GemmDesc& g, int n_chains, hipStream_t s);

// ---- elementwise.hip ----------------------------------------------------------
// expV[l][i] = tabV[f[l][i]], invexpV likewise (AttractiveHubbard::expV/invexpV)
int launch_build_expv(const int8_t* fields, long f_stride, int nt, int n, const double* tab_expv /*[chains][8]*/,
                      double* expv, double* invexpv, long v_stride, int n_chains, hipStream_t s);
// split d into 1/max(d,1) and min(d,1) (source/stablelinalg.cpp:100-108); also log-sum of max(d,1) into *logsum
int launch_split_d(CVec d, Vec dl_inv, Vec ds, double* logsum, int n, int n_chains, hipStream_t s);
int launch_split_d2(CVec d1, Vec dl1, Vec ds1, CVec d2, Vec dl2, Vec ds2, int n, int n_chains, hipStream_t s);   // two splits, one launch
// equal-time observables of G (source/model.cpp:167-288) in displacement space (include/measurementh5.h:13-66):
// out[chain][0..2] = density, doubleOcc, swave; out[chain][3 + dx_idx + L1*dy_idx] = chi_r.  accumulate: out += (else out =)
int launch_measure_equal_time(CMat G, double* out, long out_stride, int L1, int L2, int accumulate, int n_chains, hipStream_t s);
// dynamical observables (source/model.cpp:290-394) of the series Gtt / Gt0 / G0t [nt + 1][chains][n*n] in displacement space:
// out[chain][obs][tau][bin], obs = greenTau, doublonTau, currxxTau.  accumulate: out += (else out =)
int launch_measure_unequal_time(const double* Gtt, const double* Gt0, const double* G0t, double* out, long out_stride, int L1, int L2, int nt,
                                int accumulate, int n_chains, hipStream_t s);
// out = a * in + b * I (n x n)
int launch_axpb_identity(CMat in, Mat out, double a, double b, int n, int n_chains, hipStream_t s);
// out = in (n*n)
int launch_copy(CMat in, Mat out, long count, int n_chains, hipStream_t s);
// out = I
int launch_set_identity(Mat out, int n, int n_chains, hipStream_t s);
// out[i,j] = rs[i]*in^T[i,j]  (= rs[i]*in[j,i]);  rs may be null
int launch_transpose_scale(CMat in, Mat out, CVec rs, int n, int n_chains, hipStream_t s);
// err[c] = max|A-B|
int launch_max_abs_diff(CMat A, CMat B, double* err, long err_stride, int n, int n_chains, hipStream_t s);
// C = A + B * diag(cs)   (cs may be null)
int launch_add_scaled_cols(CMat A, CMat B, CVec cs, Mat C, int n, int n_chains, hipStream_t s);
// out = diag(rs) * in
int launch_scale_rows(CMat in, CVec rs, Mat out, int n, int n_chains, hipStream_t s);
// fold one half-sweep's per-slice accepted counts and per-stabilisation errors into the per-chain stats
struct DevStats { double acc_rate, max_err, sum_err, n_err; long long n_accepted, n_proposed; };
int launch_fold_stats(DevStats* st, const int* acc, long acc_stride, int n_slices, const double* err, long err_stride, int n_err,
                      int n, int nt, int n_chains, hipStream_t s);

// ---- qr.hip -----------------------------------------------------------------
// to_LDR on device: A (overwritten with reflectors / R0), outputs L (=Q), d, R.
// work: per chain scratch of at least 4*n doubles + n ints (tau, norms, perm).
struct QrWork {
    double* tau; long tau_stride;      // n per chain
    int* jpvt; long jpvt_stride;       // n per chain
    unsigned long long* sync = nullptr; long sync_stride = 0;   // cooperative QRCP: granule records, >= qrcp_coop_sync_granules(n) per chain
    int* abort_words = nullptr;                                 // cooperative QRCP: one word per chain
    int* info = nullptr;                                        // |= 2 when a cooperative factorisation gave up waiting
    double* pw = nullptr; long pw_stride = 0;                   // panel-pivoted QR (qr_panel.hip): sketch, clean reflector panel, T; >= qr_panel_work_doubles(n) per chain
    int* pivpos = nullptr; long pivpos_stride = 0;              // panel-pivoted QR: pivot position of every column (-1 = live), n per chain
};
long qr_panel_work_doubles(int n);
bool qr_panel_ok(int n, const QrWork& w);   // n a multiple of 16 in [16, 1024] and the workspace present
int launch_qr_panel_formq(QrWork w, Mat L, int n, int n_chains, hipStream_t s);   // explicit Q of that factorisation from its compact-WY factors (left in w.pw)
int launch_qr_panel(Mat A, QrWork w, int n, int n_chains, hipStream_t s);   // qr_panel.hip: A -> reflectors / R0 in place, tau, jpvt (same format as the QRCP kernels)
long qrcp_coop_sync_granules(int n);        // granules of cooperative-QRCP workspace per chain
int qrcp_coop_workgroups(int n, int n_chains);
int launch_to_ldr(Mat A, Mat L, Vec d, Mat R, QrWork w, int n, int n_chains, hipStream_t s);

// ---- lu.hip -----------------------------------------------------------------
// In-place LU with partial pivoting of A[c] (P A = L U); perm (n ints per
// chain): row r of P*A is row perm[r] of A.  logabsdet[c] (+)= sum log|u_ii|
// when non-null.  *info |= 1 on a zero / NaN pivot.
// rowpos: scratch of n ints per chain for the blocked path (null -> single-workgroup streaming kernel)
int launch_lu(Mat A, int* perm, long perm_stride, double* logabsdet, int accumulate_logdet, int* info, int n, int n_chains, hipStream_t s,
              int* rowpos = nullptr, long rowpos_stride = 0);
// Solve A X = B with the factors above, n right-hand sides.
//   mode 0: X holds B on entry, overwritten by the solution.
//   mode 1: B = diag(dg); X is overwritten.
//   mode 2: "LU" is the R of a to_LDR result and perm its jpvt: X = R^-1 diag(dg) by a permuted
//           triangular back substitution (no factorisation needed).
int launch_lu_solve(CMat LU, const int* perm, long perm_stride, Mat X, CVec dg, int mode, int n, int n_chains, hipStream_t s);

// ---- tri_solve.hip ------------------------------------------------------------
// X = R^-1 diag(dg) for the permuted-triangular R of one to_LDR (perm = its jpvt), n <= 640: blocked back substitution on
// the matrix cores (n/16 dependent steps instead of n).  scratch: 16 * (n + 16) doubles per chain (inverses of the diagonal blocks).
int launch_tri_solve(CMat R, const int* perm, long perm_stride, Mat X, CVec dg, double* scratch, long scratch_stride, int n, int n_chains, hipStream_t s);

// ---- lu_gj.hip ----------------------------------------------------------------
// X = A^-1 B for n <= 256 by blocked Gauss-Jordan elimination with partial pivoting (no substitution phase); A and B
// are destroyed.  SA: n*n scratch per chain, tinv: 2048 doubles per chain, perm / rowpos: n ints per chain.
// logabsdet (optional): (+)= log|det A|.  *info |= 1 on a zero / NaN pivot.
int launch_gj_solve(Mat A, Mat B, Mat X, Mat SA, double* tinv, int* perm, long perm_stride, int* rowpos, long rowpos_stride,
                    double* logabsdet, int accumulate_logdet, int* info, int n, int n_chains, hipStream_t s);

// ---- update.hip ---------------------------------------------------------------
struct UpdateTables {            // per chain, 64 doubles: model constants the slice kernel needs
    // rb[old][k]   = (gamma[new]/gamma[old]) * exp(alpha*g*(eta[new]-eta[old]))   (source/model.cpp:99-122)
    // delta[old][k]= 1/bosonR - 1
    // ev[f] = exp(g*eta[f]), iev[f] = exp(-g*eta[f])
    double rb[4][3];
    double delta[4][3];
    double ev[4];
    double iev[4];
    double pad[32];
};
struct UpdateDesc {
    Mat G;                        // current Green's function (n*n per chain)
    int8_t* fields; long f_stride;            // [chain][nt][n]
    double* expv; double* invexpv; long v_stride;   // [chain][nt][n]
    const UpdateTables* tabs;                 // [chain]
    const int32_t* perm; const uint8_t* kprop; const double* u; long rs_stride;  // random stream [chain][nt][n]
    double* Upanel; double* Wpanel; long panel_stride;   // [chain][KD][n]
    double* Cpanel = nullptr;                 // [chain][KD][KD]: the window's k x k matrix of the sub-matrix walk (update_sm.hip); null -> delayed-update walk
    int* state; long state_stride;            // per chain: [0]=pos, [1]=k (accepts in current window), [2]=accepted in slice
    double* prep; long prep_stride;           // per chain 4*n doubles: the slice's G-independent proposal data
    Mat GT{nullptr, 0};                       // transposed copy of G the n <= 256 walk reads rows from; kept in step by the flushes
    int gt_valid = 0;                         // GT already equals G^T (written by the GEMM that produced G): skip the transpose launch
    void* slice_sync = nullptr;               // per chain SLICE_SYNC_BYTES: hand-off words of the persistent slice kernels (null -> multi-kernel path)
    unsigned slice_epoch = 0;                 // launch number of the persistent slice kernel on this engine (1, 2, ...): every hand-off word carries it
    int slice_absent_tile = -1;               // debug (DQMC_DEBUG_SLICE_ABSENT): this flush workgroup leaves at once, as if it had never become resident
    int slice_absent_l = -1;                  // debug (DQMC_DEBUG_SLICE_ABSENT=<tile>:<slice>): only in the launch of that time slice (-1: in every launch)
    int slice_late_tile = -1, slice_late_us = 0;   // debug (DQMC_DEBUG_SLICE_LATE=<tile>:<us>): this flush workgroup sleeps that long before it checks in
    int* info = nullptr;                      // |= 4 when a hand-off of the persistent slice kernel timed out
    int* acc_out; long acc_stride;            // per chain per slice accepted counts [chain][2*nt] (+ offset chosen by caller)
    int n, nt;
};
constexpr int UPDATE_KD = 32;    // delayed-update window (accepted flips per flush)

// ---- hand-off words of the persistent slice kernels (update.hip: slice_kernel, update_sm.hip: slice_sm_kernel) ----------------
// One block per chain.  Every word a workgroup polls carries the LAUNCH number next to the window number:
//   tag = (epoch << 8) | window,   epoch = 1, 2, ... per launch on the engine (UpdateDesc::slice_epoch), window = 0 .. 255
// so a word left behind by an earlier launch can never match and nothing has to be re-armed when a launch ends (the round-2
// kernels zeroed the words through an exit ticket: 66 stores and one atomic per workgroup on the tail of every launch).  The host
// zeroes the block and restarts the epoch before it reaches 2^24 (Engine::local_update, engine.hip).
//   seq       walk -> flush: high word = tag of the window just closed, low word = k | solo << 30 | final << 31
//   arrive[f] flush -> walk: tag of the last window flush workgroup f has absorbed; window 0 = "resident" (the census)
struct SliceSync {
    unsigned long long seq; unsigned int error; unsigned int solo_count; unsigned int pad[12];    // 64 B
    unsigned int arrive[496];
};
constexpr size_t SLICE_SYNC_BYTES = 2048;
static_assert(sizeof(SliceSync) == SLICE_SYNC_BYTES, "SliceSync layout");
constexpr unsigned SLICE_EPOCH_LIMIT = 1u << 24;
constexpr unsigned SLICE_SOLO_BIT = 1u << 30, SLICE_FINAL_BIT = 1u << 31;
__host__ __device__ inline unsigned slice_tag(unsigned epoch, unsigned window) { return (epoch << 8) | window; }
// one slice = reset + windows x (scan kernel, flush kernel)
// *gt_kept (optional): 1 when the path taken keeps d.GT equal to G^T, 0 when it leaves GT behind (n > 256 on the scan / flush kernel pairs)
int launch_update_slice(const UpdateDesc& d, int l, int acc_slot, int n_chains, hipStream_t s, int* gt_kept = nullptr);
// sub-matrix variant of the persistent single-launch slice kernel (update_sm.hip)
int launch_update_slice_sm(const UpdateDesc& d, int l, int acc_slot, int n_chains, hipStream_t s);
int slice_flush_workgroups(int n);        // flush workgroups per chain of the persistent slice kernels
// CU reservation for the persistent single-launch slice kernel (see update.hip); an engine that holds one passes slice_sync
bool slice_reserve(int device, int n, int n_chains);
void slice_release(int device, int n, int n_chains);
// standalone Sherman-Morrison rank-1 update (source/model.cpp:124-138), whole-chip streaming kernel
int launch_rank1(Mat G, int i, double delta, double* scratch /*2n+1 doubles per chain*/, long scratch_stride, int n, int n_chains, hipStream_t s);

// ---- checkerboard.hip -----------------------------------------------------------
// out = diag(rs_out) * [ E^(+-1) * (diag(rs_in) * in * diag(cs_in)) ] * diag(cs_out), and / or its transpose into outT, where
// E = f * E_{G-1} ... E_0 is the checkerboard break-up of exp(-dtau K) (README.md:40): E_g mixes the site pairs (r, partner[g][r])
// with [cosh sinh; sinh cosh].  reverse: run the groups G-1 .. 0 instead of 0 .. G-1; inverse: sinh -> -sinh, f -> 1/f.
// E^-1 * M is (reverse, inverse); M * E^(+-1) is the same launch on M^T (E_g symmetric).  `out` may alias `in`, `outT` may not.
struct CbDesc {
    CMat in;
    Mat out{nullptr, 0}, outT{nullptr, 0};
    const int* partner = nullptr;             // [n_groups][n]; partner[g][r] == r: site r is in no pair of group g
    int n_groups = 0, reverse = 0, inverse = 0;
    const double* par = nullptr; long par_stride = 0;   // per chain {cosh, sinh, f, 1/f}
    CVec rs_in, cs_in, rs_out, cs_out;        // null = identity
    int n = 0;
};
int launch_cb_apply(const CbDesc& d, int n_chains, hipStream_t s);

// ---- per-device kernel attributes (dynamic LDS > 64 KiB), set once per device by init_device_kernels (engine.hip) ----
int update_init_device();
int update_sm_init_device();
int qr_init_device();
int qr_colown_init_device();
int init_device_kernels(int device);

// ---- engine.hip internals replica.hip needs: the HBM-resident HS fields of a single-chain engine ----
struct EngineFieldsView { int device, n, nt, n_chains; int8_t* fields; hipStream_t stream; };
int engine_fields_view(dqmc_engine* h, EngineFieldsView* v);
// the device fields were overwritten in place: rebuild the exp(+-g eta) tables, the stack and G are stale until dqmc_init
int engine_fields_changed(dqmc_engine* h);

}  // namespace dq
