// host_capi.cpp -- C entry points onto the C++ host facade (dqmc_host.hpp) so the
// CPU test-suite can exercise the host logic (RNG stream semantics, INI reader,
// model constants, replica pairing) without a GPU.  No HIP, no oracle.
#include "dqmc_host.hpp"
#include "results_h5.hpp"

#include <cstdio>
#include <memory>
#include <thread>

extern "C" {

int dqmc_host_abi_version(void) { return 1; }

void* dqmc_host_rng_create(unsigned int seed) { return new utility::random(seed); }
void dqmc_host_rng_destroy(void* r) { delete static_cast<utility::random*>(r); }
unsigned int dqmc_host_rng_next(void* r) { return static_cast<utility::random*>(r)->get_generator()(); }

// the production path: update::draw_slice_stream
void dqmc_host_draw_slice(void* r, int nv, int32_t* perm, uint8_t* kprop, double* u) {
    update::draw_slice_stream(*static_cast<utility::random*>(r), nv, perm, kprop, u);
}

// a literal restatement of what update::local_update draws (source/update.cpp:10-25), using the
// reference's own constructs: std::shuffle, GHQField::propose_new_field with the generator passed
// BY VALUE, and the canonical uniform std::bernoulli_distribution compares with p.
void dqmc_host_draw_slice_literal(void* r, int nv, int32_t* perm, uint8_t* kprop, double* u) {
    utility::random& rng = *static_cast<utility::random*>(r);
    std::vector<int> field_order(nv);
    for (int i = 0; i < nv; ++i) field_order[i] = i;
    std::shuffle(field_order.begin(), field_order.end(), rng.get_generator());
    GHQField fld;   // tables only
    for (int idx = 0; idx < nv; ++idx) {
        perm[idx] = field_order[idx];
        const int proposed = fld.propose_new_field(0, rng);          // old = 0: proposal row {1,2,3}
        kprop[idx] = (uint8_t)(proposed - 1);
        u[idx] = std::generate_canonical<double, std::numeric_limits<double>::digits>(rng.get_generator());
    }
}

// bernoulli(p) of include/utility.h:34-37 next to "u < p" on a twin generator: returns how many of n trials differ
int dqmc_host_bernoulli_check(unsigned int seed, int n, const double* p) {
    utility::random a(seed), b(seed); int diff = 0;
    for (int k = 0; k < n; ++k) {
        const bool x = a.bernoulli(p[k]);
        const double c = std::generate_canonical<double, 53>(b.get_generator());
        if (x != (c < p[k])) ++diff;
    }
    if (a.get_generator()() != b.get_generator()()) ++diff;
    return diff;
}

int dqmc_host_partner_rank(int rank, int world, int attempt) { return update::partner_rank(rank, world, attempt); }

void dqmc_host_expm(int n, const double* A, double* out) {
    std::vector<double> a(A, A + (size_t)n * n);
    const std::vector<double> e = dqmc_detail::expm(a, n);
    std::copy(e.begin(), e.end(), out);
}

// builds Lattice + AttractiveHubbard from an INI text and returns its constants; returns 0, or -1 with *err filled
int dqmc_host_model(const char* ini, double beta, unsigned int seed, int* ns, int* nt, double* g, double* expK, double* invexpK,
                    int64_t* fields, double* gamma, double* eta, char* err, int errlen) {
    try {
        utility::parameters params = utility::parameters::from_string(ini);
        utility::random rng(seed);
        Lattice lat(params, {1.0, 0.0}, {0.0, 1.0}, {{0.0, 0.0}});
        AttractiveHubbard model(params, lat, rng, beta);
        *ns = model.ns(); *nt = model.nt(); *g = model.g();
        const size_t nn = (size_t)model.ns() * model.ns();
        if (expK) std::copy(model.expK(0).begin(), model.expK(0).end(), expK);
        if (invexpK) std::copy(model.invexpK(0).begin(), model.invexpK(0).end(), invexpK);
        (void)nn;
        if (fields) std::copy(model.fields().fields().begin(), model.fields().fields().end(), fields);
        for (int k = 0; k < 4; ++k) { if (gamma) gamma[k] = model.fields().gamma(k); if (eta) eta[k] = model.fields().eta(k); }
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}

// AttractiveHubbard::checkerboard_groups of the INI's lattice: bonds[2 * max_bonds] group after group, sizes[max_groups], par = {cosh, sinh, exp(dtau mu)};
// returns the number of groups, or -1 with *err filled
int dqmc_host_checkerboard_groups(const char* ini, double beta, int32_t* bonds, int max_bonds, int32_t* sizes, int max_groups, double* par, char* err, int errlen) {
    try {
        utility::parameters params = utility::parameters::from_string(ini);
        utility::random rng(1);
        Lattice lat(params, {1.0, 0.0}, {0.0, 1.0}, {{0.0, 0.0}});
        AttractiveHubbard model(params, lat, rng, beta);
        const auto groups = model.checkerboard_groups();
        if ((int)groups.size() > max_groups) throw std::length_error("too many groups");
        int b = 0;
        for (size_t g = 0; g < groups.size(); ++g) {
            sizes[g] = (int32_t)groups[g].size();
            for (const auto& q : groups[g]) { if (b >= max_bonds) throw std::length_error("too many bonds"); bonds[2 * b] = q[0]; bonds[2 * b + 1] = q[1]; ++b; }
        }
        if (par) { par[0] = model.cosh_dtau_t(); par[1] = model.sinh_dtau_t(); par[2] = model.exp_dtau_mu(); }
        return (int)groups.size();
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}

// INI reader probes: kind 0 = int, 1 = double, 2 = bool, 3 = length of double list
int dqmc_host_param(const char* ini, const char* section, const char* key, int kind, double* out, char* err, int errlen) {
    try {
        utility::parameters p = utility::parameters::from_string(ini);
        if (kind == 0) *out = p.getInt(section, key);
        else if (kind == 1) *out = p.getDouble(section, key);
        else if (kind == 2) *out = p.getBool(section, key) ? 1.0 : 0.0;
        else *out = (double)p.getDoubleVector(section, key).size();
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}

// ---- results_h5.hpp: the on-disk layout of the reference (include/measurementh5.h:277-362) ----
// Writes n_bins bins into <dir>/data_<rank>.h5.  scalars: [n_bins][3] (density, doubleOcc, swave); chi_r: [n_bins][L1*L2];
// unequal: NULL or [n_bins][3][n_tau][L1*L2].  The lattice (a1, a2, k-points) is the square one the driver builds.
int dqmc_host_results_write(const char* dir, int rank, int L1, int L2, int n_bins, const double* scalars, const double* chi_r,
                            const double* unequal, int n_tau, char* err, int errlen) {
    try {
        utility::parameters params = utility::parameters::from_string("[Lattice]\nL1 = " + std::to_string(L1) + "\nL2 = " + std::to_string(L2) + "\n");
        Lattice lat(params, {1.0, 0.0}, {0.0, 1.0}, {{0.0, 0.0}});
        ResultsWriter w(dir, rank, L1, L2, lat.a1(), lat.a2(), lat.k_points());
        const size_t nb = (size_t)L1 * L2;
        for (int b = 0; b < n_bins; ++b) {
            ResultsBin bin;
            bin.density = scalars[3 * b]; bin.doubleOcc = scalars[3 * b + 1]; bin.swave = scalars[3 * b + 2];
            bin.densityCorr_r.assign(chi_r + b * nb, chi_r + (b + 1) * nb);
            if (unequal) { bin.n_tau = n_tau; bin.unequal_r.assign(unequal + (size_t)b * 3 * n_tau * nb, unequal + (size_t)(b + 1) * 3 * n_tau * nb); }
            w.write_bin(bin);
        }
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}
// Reads a fp64 dataset back: ndims / dims (up to 8) / data (up to capacity doubles).  Returns the element count, or -1.
long long dqmc_host_results_read(const char* file, const char* dataset, int* ndims, unsigned long long* dims, double* data, long long capacity,
                                 char* err, int errlen) {
    try {
        std::vector<hdf5::hsize_t> d;
        const std::vector<double> v = hdf5::read_array(file, dataset, &d);
        if (ndims) *ndims = (int)d.size();
        if (dims) for (size_t k = 0; k < d.size() && k < 8; ++k) dims[k] = d[k];
        if (data) std::copy(v.begin(), v.begin() + std::min<long long>((long long)v.size(), capacity), data);
        return (long long)v.size();
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}

}  // extern "C"

// ---- parallel tempering with in-process replicas (threads) -------------------------------------------------------
// What source/main.cpp builds per MPI rank (rng, model at the rank's beta, DQMC, stacks, greens) for `world` replicas
// of ONE process, their engines on the given devices, meeting in an update::InProcessHub.  The test-suite drives
// update::replica_exchange through this on HIP engines (tests/test_gpu_parity.py) and checks it against CPU-oracle
// engines fed with the same random streams; the driver's single-process PT mode is the same construction.
namespace {
struct PtReplica {
    utility::random rng; AttractiveHubbard model; DQMC sim;
    std::vector<LDRStack> stacks; std::vector<GF> greens; dqmc_comm* comm = nullptr;
    int attempt = 0, accepted = 0;
    PtReplica(const utility::parameters& p, const Lattice& lat, unsigned seed, double beta, int device)
        : rng(seed), model(p, lat, rng, beta), sim(p, model, device), stacks(1), greens(1) {
        stacks[0] = sim.init_stacks(0); greens[0] = sim.init_greenfunctions(stacks[0]);
    }
    ~PtReplica() { dqmc_comm_destroy(comm); }
};
struct PtWorld {
    utility::parameters params; std::unique_ptr<Lattice> lat; std::unique_ptr<update::InProcessHub> hub;
    std::vector<std::unique_ptr<PtReplica>> reps;
    template <class F> int each(F&& f, char* err, int errlen) {
        std::vector<std::thread> th; std::vector<std::string> errs(reps.size());
        for (size_t r = 0; r < reps.size(); ++r) th.emplace_back([&, r] { try { f((int)r, *reps[r]); } catch (const std::exception& e) { errs[r] = e.what(); if (errs[r].empty()) errs[r] = "error"; hub->abort(); } });
        for (auto& t : th) t.join();
        for (size_t r = 0; r < reps.size(); ++r) if (!errs[r].empty()) {
            if (err && errlen > 0) { std::snprintf(err, (size_t)errlen, "replica %zu: %s", r, errs[r].c_str()); }
            return -1;
        }
        return 0;
    }
};
}  // namespace

extern "C" {

void* dqmc_host_pt_create(const char* ini, int world, const double* betas, const unsigned int* seeds, const int* devices, char* err, int errlen) {
    try {
        std::unique_ptr<PtWorld> w(new PtWorld);
        w->params = utility::parameters::from_string(ini);
        w->lat.reset(new Lattice(w->params, {1.0, 0.0}, {0.0, 1.0}, {{0.0, 0.0}}));
        w->hub.reset(new update::InProcessHub(world));
        for (int r = 0; r < world; ++r) {
            w->reps.emplace_back(new PtReplica(w->params, *w->lat, seeds[r], betas[r], devices ? devices[r] : 0));
            w->reps.back()->comm = w->hub->make_comm(r);
        }
        return w.release();
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return nullptr;
    }
}
void dqmc_host_pt_destroy(void* p) { delete static_cast<PtWorld*>(p); }

// model.fields().set_fields + init_stacks + init_greenfunctions of one replica
int dqmc_host_pt_set_fields(void* p, int rank, const int64_t* fields, char* err, int errlen) {
    try {
        PtReplica& r = *static_cast<PtWorld*>(p)->reps.at(rank);
        std::vector<int64_t> f(fields, fields + (size_t)r.model.nt() * r.model.ns());
        r.model.fields().set_fields(f);
        r.stacks[0] = r.sim.init_stacks(0); r.greens[0] = r.sim.init_greenfunctions(r.stacks[0]);
        return 0;
    } catch (const std::exception& e) { if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return -1; }
}
// n_sweeps x (sweep_0_to_beta + sweep_beta_to_0) on every replica (concurrently = 1: the replicas' threads run at once, as ranks do)
int dqmc_host_pt_sweeps(void* p, int n_sweeps, int concurrently, char* err, int errlen) {
    PtWorld& w = *static_cast<PtWorld*>(p);
    auto body = [&](int, PtReplica& r) {
        for (int i = 0; i < n_sweeps; ++i) { r.sim.sweep_0_to_beta(r.greens, r.stacks); r.sim.sweep_beta_to_0(r.greens, r.stacks); }
        r.sim.download(r.greens[0]);
    };
    if (concurrently) return w.each(body, err, errlen);
    try { for (size_t r = 0; r < w.reps.size(); ++r) body((int)r, *w.reps[r]); return 0; }
    catch (const std::exception& e) { if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return -1; }
}
// one round of update::replica_exchange on every replica (source/main.cpp:146-153: barrier, then the exchange); results [world]
int dqmc_host_pt_exchange(void* p, dqmc_exchange_result* results, char* err, int errlen) {
    PtWorld& w = *static_cast<PtWorld*>(p);
    const int world = (int)w.reps.size();
    return w.each([&](int rank, PtReplica& r) {
        dqmc_detail::check(dqmc_comm_barrier(r.comm));
        results[rank] = update::replica_exchange(rank, world, r.rng, r.attempt, r.accepted, r.model, r.sim, r.greens, r.stacks, r.comm);
    }, err, errlen);
}
// state of one replica: model.fields() (nt x nv column-major int64), greens[0].Gtt[0], log_det_M, S = global_action, counters
int dqmc_host_pt_get(void* p, int rank, int64_t* fields, double* G, double* logdet, double* action, int* attempt, int* accepted, char* err, int errlen) {
    try {
        PtReplica& r = *static_cast<PtWorld*>(p)->reps.at(rank);
        r.sim.download_fields(); r.sim.download(r.greens[0]);
        if (fields) std::copy(r.model.fields().fields().begin(), r.model.fields().fields().end(), fields);
        if (G) std::copy(r.greens[0].Gtt0.begin(), r.greens[0].Gtt0.end(), G);
        if (logdet) *logdet = r.greens[0].log_det_M;
        if (action) *action = r.sim.global_action();
        if (attempt) *attempt = r.attempt;
        if (accepted) *accepted = r.accepted;
        return 0;
    } catch (const std::exception& e) { if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return -1; }
}
// DQMC::max_err() of one replica (the largest wrap-vs-stabilised difference since its engine was created)
double dqmc_host_pt_max_err(void* p, int rank) { try { return static_cast<PtWorld*>(p)->reps.at(rank)->sim.max_err(); } catch (...) { return -1.0; } }
// dqmc_debug_snapshot + dqmc_slice_path of one replica's engine (scripts/pt_stress.py dumps them when two worlds disagree)
int dqmc_host_pt_debug(void* p, int rank, double* wrap_err, int* accepted, unsigned int* sync_words, unsigned int* slice_epoch, int* slice_path) {
    try {
        dqmc_engine* e = static_cast<PtWorld*>(p)->reps.at(rank)->sim.handle();
        if (slice_path) *slice_path = dqmc_slice_path(e);
        return dqmc_debug_snapshot(e, wrap_err, accepted, sync_words, slice_epoch);
    } catch (...) { return -1; }
}
// the raw generator of a replica: lets a test advance a twin utility::random in lock-step
unsigned int dqmc_host_pt_rng_peek(void* p, int rank) { std::mt19937 g = static_cast<PtWorld*>(p)->reps.at(rank)->rng.get_generator(); return g(); }

// the canonical uniform bernoulli(p) compares with p (update::draw_bernoulli_uniform), for twins of a rank's generator
double dqmc_host_rng_bernoulli_uniform(void* r) { return update::draw_bernoulli_uniform(*static_cast<utility::random*>(r)); }
int dqmc_host_rng_bernoulli(void* r, double p) { return static_cast<utility::random*>(r)->bernoulli(p) ? 1 : 0; }

// the in-process hub alone (no engine): `world` threads each exchange a few messages; returns 0 when every byte arrived
int dqmc_host_hub_selftest(int world, int rounds) {
    update::InProcessHub hub(world);
    std::vector<int> bad(world, 0); std::vector<std::thread> th;
    for (int r = 0; r < world; ++r) th.emplace_back([&, r] {
        for (int a = 1; a <= rounds; ++a) {
            const int partner = update::partner_rank(r, world, a);
            std::vector<int64_t> s(1000 + a, (int64_t)r * 1000 + a), g(1000 + a, -1);
            if (update::InProcessHub::sendrecv(hub.endpoint(r), s.data(), g.data(), s.size() * sizeof(int64_t), partner, 0) != 0) { bad[r]++; continue; }
            for (int64_t v : g) if (v != (int64_t)partner * 1000 + a) { bad[r]++; break; }
            double x = r + 0.5, y = -1; 
            if (update::InProcessHub::sendrecv(hub.endpoint(r), &x, &y, sizeof(double), partner, 1) != 0 || y != partner + 0.5) bad[r]++;
        }
    });
    for (auto& t : th) t.join();
    int total = 0; for (int b : bad) total += b;
    return total;
}

}  // extern "C"
