// host_capi.cpp -- C entry points onto the C++ host facade (dqmc_host.hpp) so the
// CPU test-suite can exercise the host logic (RNG stream semantics, INI reader,
// model constants, replica pairing) without a GPU.  No HIP, no oracle.
#include "dqmc_host.hpp"

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

}  // extern "C"
