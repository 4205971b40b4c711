/*
 * dqmc_hip.h -- C ABI of the MI355X-native DQMC equal-time sweep engine.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json
 * (SURVEY.md section 8): the equal-time Green's-function sweep of kfkq/DQMC.
 * The reference has no FFI layer (everything is statically linked C++,
 * CMakeLists.txt:44-53); the entry points below are what a C-ABI binding of
 * its source-level API for this path would bind.  Each declaration cites the
 * reference interface it replaces.  INTEGRATION.md shows the reference-side
 * shim (a replacement source/dqmc.cpp + source/update.cpp body that forwards
 * to these symbols).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - all matrices are column-major fp64 with leading dimension n (Armadillo
 *     layout, `arma::mat::memptr()`), all host pointers unless stated.
 *   - HS fields cross the boundary exactly as `arma::imat::memptr()` gives
 *     them: nt x nv, column-major, 64-bit signed (element (l,i) at l + nt*i;
 *     include/field.h:21, source/update.cpp:60-69).
 *   - every function returns 0 on success, a negative DQMC_E* code on
 *     failure; dqmc_last_error() gives the message for the calling thread.
 *     The reference throws std::runtime_error / std::out_of_range at the
 *     same places (source/stablelinalg.cpp:43-45, include/stackngf.h:61-75);
 *     the C++ facade in dqmc_amd/host rethrows.
 *   - the library REQUIRES a gfx950 GPU: there is no CPU fallback.  Without
 *     a device dqmc_create() and every stateless compute call fail with
 *     DQMC_ENODEVICE.
 *
 * The same function set with prefix `orc_` instead of `dqmc_` is exported by
 * the CPU oracle (oracle/dqmc_oracle.cpp) so the parity tests can drive both
 * through one harness.  The oracle is test infrastructure only.
 */
#ifndef DQMC_HIP_H
#define DQMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DQMC_OK         0
#define DQMC_EINVAL    -1   /* bad argument (shape, index, null pointer)      */
#define DQMC_ENODEVICE -2   /* no usable gfx950 device / HIP runtime error    */
#define DQMC_ENUMERIC  -3   /* factorisation broke down (zero pivot / NaN)    */
#define DQMC_ERANGE    -4   /* stack / slice index out of range               */
#define DQMC_ENOMEM    -5

typedef struct dqmc_engine dqmc_engine;   /* opaque; one Markov chain on one GPU */

/* Per-engine counters.  acc_rate follows DQMC::acc_rate() (include/dqmc.h:78,
 * source/dqmc.cpp:366,424: sum over slices of accepted/N/nt, i.e. it grows by
 * ~a per half-sweep); max_err / mean_err follow DQMC::max_err()/mean_err()
 * (include/dqmc.h:79-80, source/dqmc.cpp:317-329).                           */
typedef struct dqmc_stats {
    double  acc_rate;        /* accumulated like DQMC::acc_rate_               */
    double  max_err;         /* max wrap-vs-stabilised |dG| since creation     */
    double  sum_err;         /* total_precision_error_                          */
    double  n_err;           /* num_cumulated_precision_error_                  */
    int64_t n_accepted;      /* accepted single-site proposals since creation  */
    int64_t n_proposed;      /* proposals since creation                        */
} dqmc_stats;

const char* dqmc_last_error(void);
/* "hip:gfx950" for the product library, "cpu-oracle:<blas>" for the oracle.  */
const char* dqmc_backend(void);
/* number of visible HIP devices (0 when there is none; never fails).          */
int dqmc_device_count(void);

/* ------------------------------------------------------------------------ *
 * Stateless stable linear algebra -- include/stablelinalg.h:36-46.
 * An LDR triple is (L: n*n, d: n, R: n*n).  Host pointers in and out; the
 * library stages them through HBM on `device` 0 and runs the same kernels the
 * engine uses.  These exist for parity tests and for callers that use
 * stablelinalg:: directly.
 * ------------------------------------------------------------------------ */

/* stablelinalg::to_LDR (source/stablelinalg.cpp:35-55): pivoted Householder
 * QR, M P = Q R0; d = |diag R0|; R = diag(1/d) R0 P^T; L = Q.  For n a
 * multiple of 16 in [64, 1024] P is chosen one 16-column PANEL at a time
 * (randomised panel pivoting, qr_panel.hip) instead of one column at a time
 * (dgeqp3): L is orthogonal and L diag(d) R = M to rounding as before, every
 * row of R still has its unit entry in the pivot's column, but |R_ij| <= ~1.1
 * and d is graded to a factor ~2 instead of exactly -- what the reference's
 * consumers (inv_I_plus_ldr*, the stack) rely on is the product and the
 * grading.  Other sizes, and every size when DQMC_QR_PANEL=0 is set, return
 * dgeqp3's own pivot order.                                                   */
int dqmc_to_ldr(int n, const double* M, double* L, double* d, double* R);

/* stablelinalg::ldr_mul_mat (source/stablelinalg.cpp:57-67): F' = F * M.     */
int dqmc_ldr_mul_mat(int n, const double* L, const double* d, const double* R,
                     const double* M, double* Lo, double* d_o, double* Ro);

/* stablelinalg::mat_mul_ldr (source/stablelinalg.cpp:69-79): F' = M * F.     */
int dqmc_mat_mul_ldr(int n, const double* M,
                     const double* L, const double* d, const double* R,
                     double* Lo, double* d_o, double* Ro);

/* stablelinalg::ldr_mul_ldr (source/stablelinalg.cpp:81-92): F' = F1 * F2.   */
int dqmc_ldr_mul_ldr(int n,
                     const double* L1, const double* d1, const double* R1,
                     const double* L2, const double* d2, const double* R2,
                     double* Lo, double* d_o, double* Ro);

/* stablelinalg::inv_I_plus_ldr (source/stablelinalg.cpp:94-126):
 * G = (I + F)^-1, *logdet = log|det(I + F)|.                                 */
int dqmc_inv_I_plus_ldr(int n, const double* L, const double* d, const double* R,
                        double* G, double* logdet);

/* stablelinalg::inv_I_plus_ldr_mul_ldr (source/stablelinalg.cpp:128-158):
 * G = (I + F1 F2)^-1.                                                        */
int dqmc_inv_I_plus_ldr_mul_ldr(int n,
                                const double* L1, const double* d1, const double* R1,
                                const double* L2, const double* d2, const double* R2,
                                double* G);

/* Dense C = op(A) op(B) on the fp64 MFMA GEMM kernel (the `*` of
 * source/dqmc.cpp:76,82,102,130,185).  trans flags: 0 = N, 1 = T.            */
int dqmc_gemm(int n, const double* A, int transA, const double* B, int transB, double* C);

/* AttractiveHubbard::update_greens_local (source/model.cpp:124-138), one
 * Sherman-Morrison rank-1 update of a standalone matrix:
 * G += delta/(1+(1-G_ii) delta) * G[:,i] (G[i,:] - e_i).                     */
int dqmc_rank1_update(int n, double* G, int i, double delta);

/* ------------------------------------------------------------------------ *
 * Engine -- class DQMC (include/dqmc.h:21-93) + the model/field state it
 * reads through AttractiveHubbard& (include/model.h:36-57) and GHQField
 * (include/field.h:61-83).  State (fields, stack of LDRs, current G) lives
 * in HBM across calls.
 * ------------------------------------------------------------------------ */

/* DQMC::DQMC (source/dqmc.cpp:5-36) + the constants AttractiveHubbard's
 * constructor derives (source/model.cpp:17-35): n_sites = Lattice::n_cells(),
 * nt, n_stab from [simulation]; g = sqrt(|U| dtau / 2); gamma/eta = GHQField
 * tables (include/field.h:32-43); expK = expmat(-dtau K), invexpK =
 * expmat(+dtau K) (n*n column-major).  `device` is the HIP device ordinal.   */
int dqmc_create(dqmc_engine** out, int device, int n_sites, int nt, int n_stab,
                double g, const double gamma[4], const double eta[4],
                const double* expK, const double* invexpK);
/* Batched form: one engine advancing n_chains independent Markov chains (the
 * reference's MPI ranks, source/main.cpp:20-37) on one GPU; every kernel launch
 * covers all chains.  g, expK, invexpK carry a leading n_chains dimension (each
 * chain may have its own beta, as under parallel tempering, source/main.cpp:47-67).
 * For an engine with n_chains > 1 EVERY per-chain array of the calls below
 * (fields, G, logdet, stack L/d/R, perm/kprop/u, stats, accepted, Bbar, S)
 * gains a leading n_chains dimension; dqmc_create is the n_chains = 1 case.   */
int dqmc_create_batch(dqmc_engine** out, int device, int n_chains, int n_sites, int nt, int n_stab,
                      const double* g, const double gamma[4], const double eta[4],
                      const double* expK, const double* invexpK);
int dqmc_n_chains(dqmc_engine* e);
void dqmc_destroy(dqmc_engine* e);

/* GHQField::set_fields / fields() (include/field.h:64,72-74).                */
int dqmc_set_fields(dqmc_engine* e, const int64_t* fields);
int dqmc_get_fields(dqmc_engine* e, int64_t* fields);

/* DQMC::init_stacks + DQMC::init_greenfunctions (source/dqmc.cpp:43-72):
 * rebuilds every stack[i] = B(beta, tau_i) from the current fields and sets
 * G = Gtt[0] = (I + stack[0])^-1, log_det_M.                                 */
int dqmc_init(dqmc_engine* e);

/* The equal-time Green's function the sweep currently holds (GF::Gtt[l] of
 * the slice the sweep front is at; Gtt[0] after init or a full sweep;
 * include/stackngf.h:15-29) and GF::log_det_M.                               */
int dqmc_get_G(dqmc_engine* e, double* G);
int dqmc_set_G(dqmc_engine* e, const double* G);
int dqmc_get_logdet(dqmc_engine* e, double* logdet);

/* LDRStack::operator[] (include/stackngf.h:60-67).  DQMC_ERANGE mirrors its
 * std::out_of_range.                                                         */
int dqmc_n_stack(dqmc_engine* e);
int dqmc_get_stack(dqmc_engine* e, int i_stack, double* L, double* d, double* R);

/* DQMC::sweep_0_to_beta / sweep_beta_to_0 (source/dqmc.cpp:337-456).
 * The per-slice random stream that update::local_update draws from
 * utility::random (source/update.cpp:10-25, include/field.h:76-83,
 * include/utility.h:34-37) is generated by the HOST caller and passed in,
 * indexed [l*n_sites + idx] for slice l:
 *   perm  : the std::shuffle'd site order (source/update.cpp:14)
 *   kprop : the uniform_int(0,2) proposal index (include/field.h:79-80)
 *   u     : the canonical uniform that bernoulli(p) compares with p
 * Calls are asynchronous on the engine's stream; dqmc_get_stats / dqmc_get_G
 * / dqmc_sync wait for completion.                                           */
int dqmc_sweep_0_to_beta(dqmc_engine* e, const int32_t* perm, const uint8_t* kprop, const double* u);
int dqmc_sweep_beta_to_0(dqmc_engine* e, const int32_t* perm, const uint8_t* kprop, const double* u);
int dqmc_sync(dqmc_engine* e);
int dqmc_get_stats(dqmc_engine* e, dqmc_stats* out);

/* Fine-grained steps of the sweep, for parity tests and custom drivers.      */
/* DQMC::propagate_GF_forward / _backward (source/dqmc.cpp:113-132,169-187).  */
int dqmc_wrap_forward(dqmc_engine* e, int l);
int dqmc_wrap_backward(dqmc_engine* e, int l);
/* update::local_update for slice l (source/update.cpp:5-32); perm/kprop/u
 * hold n_sites entries; *accepted receives the accepted count (synchronous). */
int dqmc_local_update_slice(dqmc_engine* e, int l, const int32_t* perm,
                            const uint8_t* kprop, const double* u, int* accepted);
/* DQMC::calculate_Bbar (source/dqmc.cpp:88-105).                             */
int dqmc_calculate_Bbar(dqmc_engine* e, int i_stack, double* Bbar);
/* AttractiveHubbard::global_action (source/model.cpp:140-159).               */
int dqmc_global_action(dqmc_engine* e, double* S);

/* ---- checkerboard break-up of the kinetic propagator ------
 * The reference multiplies by the dense exp(-+dtau K) (source/model.cpp:32-35, source/dqmc.cpp:78-132) and lists
 * "Checkerboard implementation" as future work (README.md:40).  This call switches an engine to
 *     exp(-dtau K)  ~=  E = diag_factor * E_{G-1} ... E_1 E_0,       E^-1 = E_0^-1 ... E_{G-1}^-1 / diag_factor,
 * where every E_g is a product of disjoint 2x2 blocks [cosh_t sinh_t; sinh_t cosh_t] on the site pairs of bond group g
 * (hopping K_ij = -t: cosh_t = cosh(dtau t), sinh_t = sinh(dtau t); K_ii = -mu: diag_factor = exp(dtau mu)).  The
 * engine's expK / invexpK become E / E^-1 (an O(dtau^2) Trotter break-up, a different discretisation from the
 * reference's dense exponential -- opt-in, never the default), and the wraps and B-bar products apply the factors
 * pair by pair: O(G N^2) work and 2 N^2 doubles of HBM traffic per product instead of an N^3 GEMM.
 *   bonds:       the site pairs (i, j) of all groups, group after group: [sum(group_sizes)][2]
 *   group_sizes: [n_groups]; the pairs of one group must be disjoint (sites in no pair of a group pass through)
 *   cosh_t, sinh_t, diag_factor: one value per chain (chains of a batched engine may differ in dtau)
 * Call after dqmc_create and before dqmc_init; DQMC_EINVAL on a malformed bond list.                           */
int dqmc_set_checkerboard(dqmc_engine* e, int n_groups, const int32_t* bonds, const int32_t* group_sizes,
                          const double* cosh_t, const double* sinh_t, const double* diag_factor);

/* ---- equal-time observables on the device ---------------
 * Observables::calculate_density / calculate_doubleOccupancy / calculate_swavePairing /
 * calculate_densityCorr (source/model.cpp:167-288) of the current G = Gtt[0] (what
 * measurements.measure reads after sweep_beta_to_0, source/main.cpp:165), the site
 * matrix of densityCorr reduced to displacement space exactly as
 * transform::chi_site_to_chi_r does (include/measurementh5.h:13-66, n_orb = 1):
 *   scalars [n_chains][3] = density, doubleOcc, swave
 *   chi_r   [n_chains][L1*L2], element (dx_idx, dy_idx) at dx_idx + L1*dy_idx,
 *           dx_idx = pbc_shortest(xj - xi, L1) + L1/2 - 1 (likewise dy)
 * L1*L2 must equal n_sites.  Avoids the N x N download per sweep.                */
int dqmc_measure_equal_time(dqmc_engine* e, int L1, int L2, double* scalars, double* chi_r);
/* Bin accumulation on the device (MeasurementManager::measure + accumulate,
 * include/measurementh5.h:189-274): add the observables of the current G to the
 * running sums / read sums and count back, optionally starting a new bin.        */
int dqmc_measure_accumulate(dqmc_engine* e, int L1, int L2);
int dqmc_measure_fetch(dqmc_engine* e, double* scalars_sum, double* chi_r_sum, int64_t* n_measurements, int reset);

/* ---- unequal-time path ------------------------------------
 * DQMC::sweep_unequalTime (source/dqmc.cpp:458-515; propagate_unequalTime_GF_forward
 * :223-248, propagate_Bt0_Bbt :250-264, stabilize_unequalTime :266-285, with
 * stablelinalg::inv_invldr_plus_ldr source/stablelinalg.cpp:160-190): from the current
 * G = Gtt[0], fields and stacks (as sweep_beta_to_0 leaves them) builds Gtt[l],
 * Gt0[l] = G(tau_l, 0), G0t[l] = G(0, tau_l) for l = 0..nt in HBM; the three wrap
 * errors per stabilisation go into the same statistics as check_error.  Asynchronous. */
int dqmc_sweep_unequal_time(dqmc_engine* e);
/* GF::Gtt[l] / Gt0[l] / G0t[l] (include/stackngf.h:15-29): which = 0 / 1 / 2,
 * l in 0..nt, out [n_chains][n*n] column-major.                                      */
int dqmc_get_G_tau(dqmc_engine* e, int which, int l, double* out);

/* DQMC::half_warp (source/dqmc.cpp:288-315, include/dqmc.h:87; called at source/main.cpp:161-163
 * when [simulation] symmetric = true): out = invexpK_half * M * expK_half with
 * expK_half = exp(-dtau K / 2) (AttractiveHubbard::expK_half / invexpK_half, include/model.h:41-42).
 *   which = -1: M = the current equal-time G (GF::Gtt[0]); l ignored.
 *   which = 0 / 1 / 2: M = Gtt[l] / Gt0[l] / G0t[l] of the last dqmc_sweep_unequal_time, l in 0..nt.
 * expK_half / invexpK_half: [n*n] column-major, shared by all chains; both NULL = the pair of the
 * previous call on this engine (the reference half-warps 3*nt matrices with one pair).
 * out [n_chains][n*n] column-major.  The engine's own G / series are not modified (the reference
 * writes into a second GF vector, GF_tosymm).                                                       */
int dqmc_half_warp(dqmc_engine* e, const double* expK_half, const double* invexpK_half, int which, int l, double* out);

/* Dynamical observables of the last dqmc_sweep_unequal_time in displacement space:
 * Observables::calculate_greenTau / calculate_doublonTau / calculate_currxxTau
 * (source/model.cpp:290-394) reduced per time slice as transform::chi_site_to_chi_r
 * does (include/measurementh5.h:20-66, n_orb = 1).
 *   out [n_chains][3][nt + 1][L1*L2]: observable (greenTau, doublonTau, currxxTau),
 *   slice tau, element (dx_idx, dy_idx) at dx_idx + L1*dy_idx.
 * accumulate = 0: evaluate and return (out may not be NULL); accumulate = 1: add to the
 * device-side bin sums instead (out ignored); fetch returns sums and count.           */
int dqmc_measure_unequal_time(dqmc_engine* e, int L1, int L2, int accumulate, double* out);
int dqmc_measure_unequal_fetch(dqmc_engine* e, double* out_sum, int64_t* n_measurements, int reset);

/* ---- replica exchange (parallel tempering, BASELINE.json configs[3]) -----------------
 * The reference runs one inverse temperature per MPI rank and swaps HS-field
 * configurations between neighbouring ranks with MPI_Sendrecv / MPI_Send / MPI_Recv on
 * MPI_COMM_WORLD (source/update.cpp:47-117, source/main.cpp:39-67,146-153).  Here a
 * dqmc_comm plays MPI_COMM_WORLD: one rank per GPU, point-to-point only.
 *   - RCCL transport: grouped ncclSend/ncclRecv on the engine's stream, HBM to HBM over
 *     xGMI (4 disjoint pairs per round on 8 GPUs; no ring, no all-reduce on the data path).
 *     Bootstrap as with NCCL: rank 0 obtains an id (dqmc_comm_unique_id), ships it to the
 *     other ranks by any means (file, MPI, torch.distributed store), every rank calls
 *     dqmc_comm_create_rccl.
 *   - callback transport: the caller supplies an MPI_Sendrecv-shaped function (host
 *     buffers); used for MPI itself and for in-process replicas (threads).
 * On the wire a field configuration is the engine's own int8 [nt][n_sites] array
 * (51 kB at cfg 4; the reference ships nt*nv 64-bit integers, source/update.cpp:60-69).   */
typedef struct dqmc_comm dqmc_comm;
#define DQMC_UNIQUE_ID_BYTES 128
/* blocking pairwise exchange of `bytes` bytes with rank `partner` (both sides call it with
 * the same bytes and tag); returns 0 on success.  tag: 0 fields, 1 actions, 3 decision,
 * 4.. collectives.                                                                         */
typedef int (*dqmc_sendrecv_fn)(void* user, const void* send, void* recv, size_t bytes, int partner, int tag);

int dqmc_comm_unique_id(void* id /* DQMC_UNIQUE_ID_BYTES */);
int dqmc_comm_create_rccl(dqmc_comm** out, const void* id, int world_size, int rank, int device);
int dqmc_comm_create_callbacks(dqmc_comm** out, int world_size, int rank, dqmc_sendrecv_fn fn, void* user);
void dqmc_comm_destroy(dqmc_comm* c);
int dqmc_comm_rank(dqmc_comm* c);
int dqmc_comm_world_size(dqmc_comm* c);
/* "rccl" or "callbacks" */
const char* dqmc_comm_transport(dqmc_comm* c);
/* Diagnostic: loop-back of the RCCL transport (grouped ncclSend / ncclRecv of a pattern to the own rank, two doubles
 * the same way, one ncclAllReduce); every rank of the communicator must call it.  No counterpart in the reference.   */
int dqmc_comm_selftest(dqmc_comm* c);
/* MPI_Barrier (source/main.cpp:148) and the MPI_Reduce(SUM) of source/main.cpp:186-187
 * (every rank receives the sums).                                                          */
int dqmc_comm_barrier(dqmc_comm* c);
int dqmc_comm_allreduce_sum(dqmc_comm* c, double* x, int count);

/* update::partner_rank (source/update.cpp:34-45).                                          */
int dqmc_partner_rank(int rank, int world_size, int exchange_attempt);

typedef struct dqmc_exchange_result {
    int    partner;          /* -1: no partner this attempt                                 */
    int    decider;          /* 1 when this rank drew the decision (rank < partner)          */
    int    accepted;         /* the pair's decision                                         */
    int    pad;
    double S;                /* S_r({s}_r)        (SC,               source/update.cpp:72)  */
    double S_prime;          /* S_r({s}_partner)  (SC_prime,         :81)                   */
    double S_partner;        /* SC_partner        (:88-90)                                  */
    double S_prime_partner;  /* SC_prime_partner  (:84-86)                                  */
    double deltaS;           /* (S' + S'_partner) - (S + S_partner), :94                    */
} dqmc_exchange_result;

/* One round of update::replica_exchange (source/update.cpp:47-117) for the single-chain
 * engine `e` of rank dqmc_comm_rank(comm).  `exchange_attempt` is the counter AFTER the
 * reference's `exchange_attempt++` (:52); `u` is the canonical uniform that
 * rng.bernoulli(p) of the deciding rank compares with p (:96, include/utility.h:34-37;
 * ignored on the rank that does not decide).  Field swap, the trial init_stacks +
 * init_greenfunctions (:75-80), the two action exchanges, the decision and the restoring
 * re-initialisation on rejection (:108-115) all happen here; on return G = Gtt[0] and the
 * stack belong to the fields the engine now holds.  Synchronous.                           */
int dqmc_replica_exchange_round(dqmc_engine* e, dqmc_comm* comm, int exchange_attempt, double u,
                                dqmc_exchange_result* result);

/* Number of accepted proposals / kernel time (ms, HIP events on the engine's
 * stream) spent inside the local-update kernels since the last call -- the
 * live measurement bench.py uses for the rank-1 roofline.                    */
int dqmc_update_kernel_time(dqmc_engine* e, double* ms, int64_t* n_launches, int64_t* n_accepted);
/* Diagnostic: 0 when the next local update of this engine takes the scan / flush kernel pairs (no CU reservation left,
 * or an earlier hand-off failure), 1 when it takes a persistent single-launch slice kernel, 2 when it does and at least
 * one earlier launch found a flush workgroup not resident in time and walked the slice solo (n <= 256) / left the slice
 * untouched and reported DQMC_ENUMERIC (n > 256).  The paths sum the low-rank corrections in different orders: G between
 * stabilisations agrees to ~1e-11, everything after a stabilisation bitwise.  No counterpart.                        */
int dqmc_slice_path(dqmc_engine* e);
/* Diagnostic snapshot of chain 0 for stress tooling (scripts/pt_stress.py: when two runs of one sweep disagree, ONE event must
 * be enough to analyse).  wrap_err[n_stack]: max|G_wrapped - G_stabilised| of every stabilisation of the LAST half sweep in
 * the order they were taken (source/dqmc.cpp:317-329 keeps only the running maximum); accepted[nt]: accepted flips of every
 * time slice of the last half sweep; sync_words[80]: the hand-off record of the persistent slice kernel as 32-bit words
 * (seq lo, seq hi, error, solo_count, 12 pad, arrive[0..63]); *slice_epoch: launches of that kernel so far.  Any pointer may be
 * NULL.  Synchronises the engine's stream.  No counterpart.                                                              */
int dqmc_debug_snapshot(dqmc_engine* e, double* wrap_err, int* accepted, unsigned int* sync_words, unsigned int* slice_epoch);
/* enable (1) / disable (0) the per-slice HIP-event timing above (default 0:
 * events serialise nothing but cost a few microseconds per slice).           */
int dqmc_set_profiling(dqmc_engine* e, int on);

#ifdef __cplusplus
}
#endif
#endif /* DQMC_HIP_H */
