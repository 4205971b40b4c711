// replica.hip -- replica exchange (parallel tempering) between single-chain engines, one rank per GPU:
// update::partner_rank / update::replica_exchange (source/update.cpp:34-117) and the MPI_Barrier / MPI_Reduce of the
// reference's driver (source/main.cpp:148,186-187) behind the C ABI of include/dqmc_hip.h.
//
// The reference's four messages per round (MPI_Sendrecv of the field array, two MPI_Sendrecv of one double, one
// MPI_Send/MPI_Recv of the decision) become point-to-point operations of a dqmc_comm:
//   rccl       grouped ncclSend / ncclRecv on the engine's stream, HBM to HBM (the field array never visits the host:
//              fields live in HBM as int8 [nt][n], 51 kB at cfg 4); the two actions travel as ONE message of two
//              doubles; pairs are disjoint, so on 8 GPUs a round is 4 concurrent single-link xGMI transfers.
//   callbacks  an MPI_Sendrecv-shaped host function supplied by the caller (MPI, or the in-process hub of
//              dqmc_host.hpp for replicas that are threads of one process); fields are staged through the host.
// The expensive part of a round is not the wire but the one or two from-scratch re-initialisations
// (init_stacks + init_greenfunctions, source/update.cpp:75-80,109-115), which run on the engine as dqmc_init does.
#include "common.h"
#include "../../include/dqmc_hip.h"

#include <rccl/rccl.h>

#include <cmath>
#include <cstring>
#include <vector>

using namespace dq;

struct dqmc_comm {
    int rank = 0, world = 1, device = 0;
    bool rccl = false;
    ncclComm_t nc = nullptr;
    hipStream_t stream = nullptr;            // collectives of the driver (barrier, allreduce); p2p of a round uses the engine's stream
    double* dbuf = nullptr;                  // device scratch: 8 doubles (send 0..3, recv 4..7)
    dqmc_sendrecv_fn fn = nullptr; void* user = nullptr;
    int8_t* f_mine = nullptr; int8_t* f_theirs = nullptr; size_t f_bytes = 0;   // device copies of the two field configurations
    std::vector<int8_t> h_send, h_recv;      // callback transport staging
};

#define DQ_NCCL(call)                                                                   \
    do {                                                                                \
        ncclResult_t _r = (call);                                                       \
        if (_r != ncclSuccess) {                                                        \
            ::dq::set_error(std::string(#call) + ": " + ncclGetErrorString(_r));        \
            return DQMC_ENODEVICE;                                                      \
        }                                                                               \
    } while (0)

static int have_device() {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { set_error("no HIP device available: this library requires a gfx950 GPU"); return DQMC_ENODEVICE; }
    return 0;
}

// pairwise exchange of device buffers on `s`
static int p2p(dqmc_comm* c, const void* send, void* recv, size_t bytes, int partner, int tag, hipStream_t s) {
    if (c->rccl && !c->nc) { set_error("communicator aborted after an earlier wire error: destroy it and create a new one"); return DQMC_EINVAL; }
    if (c->rccl) {
        DQ_NCCL(ncclGroupStart());
        DQ_NCCL(ncclSend(send, bytes, ncclInt8, partner, c->nc, s));
        DQ_NCCL(ncclRecv(recv, bytes, ncclInt8, partner, c->nc, s));
        DQ_NCCL(ncclGroupEnd());
        return 0;
    }
    c->h_send.resize(bytes); c->h_recv.resize(bytes);
    DQ_HIP(hipMemcpyAsync(c->h_send.data(), send, bytes, hipMemcpyDeviceToHost, s));
    DQ_HIP(hipStreamSynchronize(s));
    if (c->fn(c->user, c->h_send.data(), c->h_recv.data(), bytes, partner, tag) != 0) { set_error("replica exchange: the sendrecv callback failed"); return DQMC_EINVAL; }
    DQ_HIP(hipMemcpyAsync(recv, c->h_recv.data(), bytes, hipMemcpyHostToDevice, s));
    DQ_HIP(hipStreamSynchronize(s));
    return 0;
}
// pairwise exchange of a few doubles held on the host
static int p2p_host(dqmc_comm* c, const double* send, double* recv, int count, int partner, int tag, hipStream_t s) {
    if (!c->rccl) {
        if (c->fn(c->user, send, recv, sizeof(double) * count, partner, tag) != 0) { set_error("replica exchange: the sendrecv callback failed"); return DQMC_EINVAL; }
        return 0;
    }
    DQ_HIP(hipMemcpyAsync(c->dbuf, send, sizeof(double) * count, hipMemcpyHostToDevice, s));
    DQ_TRY_RC(p2p(c, c->dbuf, c->dbuf + 4, sizeof(double) * count, partner, tag, s));
    DQ_HIP(hipMemcpyAsync(recv, c->dbuf + 4, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    DQ_HIP(hipStreamSynchronize(s));
    return 0;
}

extern "C" {

int dqmc_partner_rank(int rank, int world_size, int exchange_attempt) {       // source/update.cpp:34-45
    const bool even_attempt = (exchange_attempt % 2 == 0);
    const int off = even_attempt ? ((rank % 2 == 0) ? 1 : -1) : ((rank % 2 == 0) ? -1 : 1);
    return (rank + off + world_size) % world_size;
}

int dqmc_comm_unique_id(void* id) {
    if (!id) { set_error("null id"); return DQMC_EINVAL; }
    DQ_TRY_RC(have_device());
    static_assert(sizeof(ncclUniqueId) == DQMC_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    DQ_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

int dqmc_comm_create_rccl(dqmc_comm** out, const void* id, int world_size, int rank, int device) {
    if (!out || !id || world_size < 1 || rank < 0 || rank >= world_size) { set_error("bad argument"); return DQMC_EINVAL; }
    DQ_TRY_RC(have_device());
    int count = 0; DQ_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) { set_error("device ordinal out of range"); return DQMC_EINVAL; }
    DQ_HIP(hipSetDevice(device));
    dqmc_comm* c = new (std::nothrow) dqmc_comm;
    if (!c) return DQMC_ENOMEM;
    c->rank = rank; c->world = world_size; c->device = device; c->rccl = true;
    ncclUniqueId u; std::memcpy(&u, id, sizeof(u));
    ncclResult_t r = ncclCommInitRank(&c->nc, world_size, u, rank);
    if (r != ncclSuccess) { set_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); delete c; return DQMC_ENODEVICE; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc(&c->dbuf, sizeof(double) * 8) != hipSuccess) {
        set_error("dqmc_comm_create_rccl: stream / scratch allocation failed"); dqmc_comm_destroy(c); return DQMC_ENODEVICE;
    }
    *out = c; return 0;
}

int dqmc_comm_create_callbacks(dqmc_comm** out, int world_size, int rank, dqmc_sendrecv_fn fn, void* user) {
    if (!out || !fn || world_size < 1 || rank < 0 || rank >= world_size) { set_error("bad argument"); return DQMC_EINVAL; }
    dqmc_comm* c = new (std::nothrow) dqmc_comm;
    if (!c) return DQMC_ENOMEM;
    c->rank = rank; c->world = world_size; c->rccl = false; c->fn = fn; c->user = user; c->device = -1;
    *out = c; return 0;
}

void dqmc_comm_destroy(dqmc_comm* c) {
    if (!c) return;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    if (c->f_mine) (void)hipFree(c->f_mine);
    if (c->f_theirs) (void)hipFree(c->f_theirs);
    if (c->dbuf) (void)hipFree(c->dbuf);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->nc) (void)ncclCommDestroy(c->nc);
    delete c;
}
int dqmc_comm_rank(dqmc_comm* c) { return c ? c->rank : -1; }
int dqmc_comm_world_size(dqmc_comm* c) { return c ? c->world : 0; }
const char* dqmc_comm_transport(dqmc_comm* c) { return !c ? "" : c->rccl ? "rccl" : "callbacks"; }

int dqmc_comm_allreduce_sum(dqmc_comm* c, double* x, int count) {
    if (!c || !x || count < 1 || count > 4) { set_error("allreduce_sum: 1..4 doubles"); return DQMC_EINVAL; }
    if (c->world == 1) return 0;
    if (c->rccl && !c->nc) { set_error("communicator aborted after an earlier wire error: destroy it and create a new one"); return DQMC_EINVAL; }
    if (c->rccl) {
        DQ_HIP(hipSetDevice(c->device));
        DQ_HIP(hipMemcpyAsync(c->dbuf, x, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
        DQ_NCCL(ncclAllReduce(c->dbuf, c->dbuf + 4, count, ncclDouble, ncclSum, c->nc, c->stream));
        DQ_HIP(hipMemcpyAsync(x, c->dbuf + 4, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
        DQ_HIP(hipStreamSynchronize(c->stream));
        return 0;
    }
    // callbacks: rank 0 collects (pairwise exchanges in rank order), then hands the sums back -- O(world) tiny messages
    double mine[4], got[4];
    std::memcpy(mine, x, sizeof(double) * count);
    if (c->rank == 0) {
        for (int r = 1; r < c->world; ++r) {
            if (c->fn(c->user, mine, got, sizeof(double) * count, r, 4) != 0) { set_error("allreduce: the sendrecv callback failed"); return DQMC_EINVAL; }
            for (int k = 0; k < count; ++k) x[k] += got[k];
        }
        for (int r = 1; r < c->world; ++r)
            if (c->fn(c->user, x, got, sizeof(double) * count, r, 5) != 0) { set_error("allreduce: the sendrecv callback failed"); return DQMC_EINVAL; }
    } else {
        if (c->fn(c->user, mine, got, sizeof(double) * count, 0, 4) != 0 || c->fn(c->user, mine, got, sizeof(double) * count, 0, 5) != 0) {
            set_error("allreduce: the sendrecv callback failed"); return DQMC_EINVAL;
        }
        std::memcpy(x, got, sizeof(double) * count);
    }
    return 0;
}
// Loop-back check of a communicator's transport: the rank exchanges a pattern with ITSELF through the same grouped
// send / receive path a round uses (ncclSend + ncclRecv to the own rank, or the callback), and, on RCCL, runs one
// ncclAllReduce; the only way to exercise the RCCL code path on a machine with a single GPU.
int dqmc_comm_selftest(dqmc_comm* c) {
    if (!c) { set_error("comm_selftest: null communicator"); return DQMC_EINVAL; }
    if (!c->rccl) { set_error("comm_selftest: RCCL transport only (a callback transport is checked by calling the callback)"); return DQMC_EINVAL; }
    if (!c->nc) { set_error("communicator aborted after an earlier wire error: destroy it and create a new one"); return DQMC_EINVAL; }
    DQ_HIP(hipSetDevice(c->device));
    const size_t bytes = 4096;
    int8_t* a = nullptr; int8_t* b = nullptr;
    DQ_HIP(hipMalloc(&a, bytes)); DQ_HIP(hipMalloc(&b, bytes));
    std::vector<int8_t> h(bytes), g(bytes, 0);
    for (size_t k = 0; k < bytes; ++k) h[k] = (int8_t)((k * 7 + c->rank) & 3);
    int rc = 0;
    do {
        if (hipMemcpyAsync(a, h.data(), bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipMemsetAsync(b, 0, bytes, c->stream) != hipSuccess) { set_error("comm_selftest: copy failed"); rc = DQMC_ENODEVICE; break; }
        rc = p2p(c, a, b, bytes, c->rank, 0, c->stream); if (rc) break;
        if (hipMemcpyAsync(g.data(), b, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { set_error("comm_selftest: copy back failed"); rc = DQMC_ENODEVICE; break; }
        if (std::memcmp(h.data(), g.data(), bytes) != 0) { set_error("comm_selftest: loop-back data mismatch"); rc = DQMC_ENUMERIC; break; }
        const double x[2] = {1.5 + c->rank, -2.0}; double y[2] = {0.0, 0.0};
        rc = p2p_host(c, x, y, 2, c->rank, 1, c->stream); if (rc) break;
        if (y[0] != x[0] || y[1] != x[1]) { set_error("comm_selftest: loop-back of two doubles mismatch"); rc = DQMC_ENUMERIC; break; }
        double z[2] = {1.0, (double)c->rank};
        if (hipMemcpyAsync(c->dbuf, z, sizeof(z), hipMemcpyHostToDevice, c->stream) != hipSuccess) { set_error("comm_selftest: copy failed"); rc = DQMC_ENODEVICE; break; }
        ncclResult_t r = ncclAllReduce(c->dbuf, c->dbuf + 4, 2, ncclDouble, ncclSum, c->nc, c->stream);
        if (r != ncclSuccess) { set_error(std::string("comm_selftest: ncclAllReduce: ") + ncclGetErrorString(r)); rc = DQMC_ENODEVICE; break; }
        if (hipMemcpyAsync(z, c->dbuf + 4, sizeof(z), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { set_error("comm_selftest: copy back failed"); rc = DQMC_ENODEVICE; break; }
        const double want1 = 0.5 * c->world * (c->world - 1);
        if (z[0] != (double)c->world || z[1] != want1) { set_error("comm_selftest: all-reduce result mismatch"); rc = DQMC_ENUMERIC; break; }
    } while (false);
    (void)hipFree(a); (void)hipFree(b);
    return rc;
}
int dqmc_comm_barrier(dqmc_comm* c) { double z = 0.0; return dqmc_comm_allreduce_sum(c, &z, 1); }

int dqmc_replica_exchange_round(dqmc_engine* e, dqmc_comm* c, int exchange_attempt, double u, dqmc_exchange_result* res) {
    if (!e || !c || !res) { set_error("replica exchange: null argument"); return DQMC_EINVAL; }
    std::memset(res, 0, sizeof(*res));
    res->partner = -1;
    EngineFieldsView v;
    DQ_TRY_RC(engine_fields_view(e, &v));
    if (v.n_chains != 1) { set_error("replica exchange: one chain per rank (a batched engine holds several)"); return DQMC_EINVAL; }
    if (c->rccl && c->device != v.device) { set_error("replica exchange: the communicator and the engine are on different devices"); return DQMC_EINVAL; }
    const int rank = c->rank, world = c->world;
    const int partner = dqmc_partner_rank(rank, world, exchange_attempt);
    if (partner < 0 || partner >= world || partner == rank) return 0;             // source/update.cpp:55-57
    res->partner = partner; res->decider = rank < partner ? 1 : 0;
    DQ_HIP(hipSetDevice(v.device));
    const size_t bytes = (size_t)v.nt * v.n;
    if (c->f_bytes != bytes) {
        if (c->f_mine) (void)hipFree(c->f_mine);
        if (c->f_theirs) (void)hipFree(c->f_theirs);
        c->f_mine = c->f_theirs = nullptr; c->f_bytes = 0;
        DQ_HIP(hipMalloc(&c->f_mine, bytes)); DQ_HIP(hipMalloc(&c->f_theirs, bytes)); c->f_bytes = bytes;
    }
    if (c->rccl && !c->dbuf) { set_error("replica exchange: communicator has no device scratch"); return DQMC_EINVAL; }
    hipStream_t s = v.stream;
    // A step that fails on THIS rank (a breakdown in dqmc_init on the trial fields, say) must not leave the partner blocked in its
    // next receive: local failures are remembered, the remaining messages are still exchanged and carry a status word, both ranks
    // treat the round as rejected (own fields restored) and both return an error.  A failure of the transport itself ends the
    // round at once; on the RCCL transport the communicator is aborted so that the peer's pending receive returns.
    int lrc = 0; std::string lmsg;
    auto local = [&](int rc) { if (rc != 0 && lrc == 0) { lrc = rc; lmsg = dqmc_last_error(); } return rc; };
    auto wire = [&](int rc) {
        if (rc != 0 && c->rccl && c->nc) { (void)ncclCommAbort(c->nc); c->nc = nullptr; }
        return rc;
    };
    // --- field exchange (MPI_Sendrecv tag 0, source/update.cpp:59-69) ---
    DQ_TRY_RC(dqmc_sync(e));                                                       // the sweep that precedes the round has finished
    DQ_HIP(hipMemcpyAsync(c->f_mine, v.fields, bytes, hipMemcpyDeviceToDevice, s));
    DQ_TRY_RC(wire(p2p(c, c->f_mine, c->f_theirs, bytes, partner, 0, s)));
    // --- S_r({s}_r), then the trial state on the partner's fields: S_r({s}_partner) (:72-81) ---
    local(dqmc_global_action(e, &res->S));
    DQ_HIP(hipMemcpyAsync(v.fields, c->f_theirs, bytes, hipMemcpyDeviceToDevice, s));
    if (local(engine_fields_changed(e)) == 0 && local(dqmc_init(e)) == 0) local(dqmc_global_action(e, &res->S_prime));
    // --- the cross actions (tags 1 and 2, :83-90) and this rank's status, one message of three doubles ---
    const double mine[3] = {res->S_prime, res->S, lrc ? 1.0 : 0.0};
    double theirs[3] = {0.0, 0.0, 0.0};
    DQ_TRY_RC(wire(p2p_host(c, mine, theirs, 3, partner, 1, s)));
    res->S_prime_partner = theirs[0]; res->S_partner = theirs[1];
    const bool broken = lrc != 0 || theirs[2] != 0.0;
    // --- decision by the lower rank (:92-105) ---
    double flag_mine = 0.0, flag_theirs = 0.0;
    if (rank < partner) {
        res->deltaS = (res->S_prime + res->S_prime_partner) - (res->S + res->S_partner);
        const double metropolis_p = std::fmin(1.0, std::exp(-res->deltaS));
        res->accepted = (!broken && u < metropolis_p) ? 1 : 0;                     // rng.bernoulli(p), include/utility.h:34-37
        flag_mine = res->accepted ? 1.0 : 0.0;
    }
    DQ_TRY_RC(wire(p2p_host(c, &flag_mine, &flag_theirs, 1, partner, 3, s)));
    if (rank > partner) {
        res->accepted = (!broken && flag_theirs != 0.0) ? 1 : 0;
        res->deltaS = (res->S_prime + res->S_prime_partner) - (res->S + res->S_partner);   // informational: the decider's is the one that counts
    }
    // --- rejected: restore the own fields and re-initialise (:108-115) ---
    if (!res->accepted) {
        DQ_HIP(hipMemcpyAsync(v.fields, c->f_mine, bytes, hipMemcpyDeviceToDevice, s));
        DQ_TRY_RC(engine_fields_changed(e));
        DQ_TRY_RC(dqmc_init(e));
    }
    if (lrc != 0) { set_error("replica exchange: " + lmsg + " (round treated as rejected on both ranks, own fields restored)"); return lrc; }
    if (broken) { set_error("replica exchange: the partner rank reported a failure during the round (treated as rejected, own fields restored)"); return DQMC_ENUMERIC; }
    return 0;
}

}  // extern "C"
