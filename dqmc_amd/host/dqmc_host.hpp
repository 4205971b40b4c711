// dqmc_host.hpp -- C++17 host facade over the C ABI (include/dqmc_hip.h).
//
// Mirrors the reference's source-level plugin surface for the equal-time
// sweep path so that a main.cpp-shaped driver reads the same
// (names, argument meaning, error behaviour):
//
//   utility::random / parameters      include/utility.h:19-48, :50-276
//   Lattice                           include/lattice.h:14-137
//   GHQField                          include/field.h:13-84
//   AttractiveHubbard                 include/model.h:11-58, source/model.cpp:3-159
//   GF, LDRStack                      include/stackngf.h:15-77
//   DQMC                              include/dqmc.h:21-93
//   update::local_update / partner_rank / replica_exchange   include/update.h:11-20
//
// Differences, all forced by where the state lives: matrices are plain
// column-major std::vector<double> instead of arma::mat (Armadillo is not a
// dependency); GF and LDRStack are thin handles onto HBM-resident state with
// explicit download; update::local_update's per-site work runs on the GPU
// inside DQMC::sweep_*, the host only draws the slice's random stream from
// utility::random with exactly the reference's calls (draw_slice_stream).
// Errors from the C ABI are rethrown as the exception types the reference
// throws (std::runtime_error, std::out_of_range).
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <mutex>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <tuple>
#include <vector>

#include "dqmc_hip.h"

namespace utility {

// include/utility.h:19-48
class random {
    std::mt19937 generator_;
    std::uniform_int_distribution<int> dist_GHQField_;
public:
    explicit random(unsigned int seed) : generator_(seed), dist_GHQField_(0, 3) {}
    bool bernoulli(double p) { std::bernoulli_distribution dist(p); return dist(generator_); }
    int rand_GHQField() { return dist_GHQField_(generator_); }
    std::mt19937& get_generator() { return generator_; }
};

// include/utility.h:50-276: INI reader (sections, '#'/';' comments, quotes,
// '_' digit separators, bool words, comma lists).  Same keys as examples/parameters.in.
class parameters {
    std::map<std::string, std::map<std::string, std::string>> sections_;
    static void trim(std::string& s) {
        const char* ws = " \t\r\n";
        const auto b = s.find_first_not_of(ws);
        if (b == std::string::npos) { s.clear(); return; }
        s = s.substr(b, s.find_last_not_of(ws) - b + 1);
    }
    static std::string strip_us(std::string v) { v.erase(std::remove(v.begin(), v.end(), '_'), v.end()); return v; }
    void parse(std::istream& in) {
        std::string line, section = "global";
        while (std::getline(in, line)) {
            const auto c = line.find_first_of("#;");
            if (c != std::string::npos) line.resize(c);
            trim(line);
            if (line.empty()) continue;
            if (line.front() == '[' && line.back() == ']') { section = line.substr(1, line.size() - 2); trim(section); continue; }
            const auto eq = line.find('=');
            if (eq == std::string::npos) continue;
            std::string key = line.substr(0, eq), val = line.substr(eq + 1);
            trim(key); trim(val);
            if (val.size() >= 2 && ((val.front() == '"' && val.back() == '"') || (val.front() == '\'' && val.back() == '\''))) val = val.substr(1, val.size() - 2);
            sections_[section][key] = val;
        }
    }
public:
    parameters() = default;
    explicit parameters(const std::string& filename) {
        std::ifstream f(filename);
        if (!f.is_open()) throw std::runtime_error("Failed to open parameter file: " + filename);
        parse(f);
    }
    static parameters from_string(const std::string& text) { parameters p; std::istringstream in(text); p.parse(in); return p; }
    void set(const std::string& section, const std::string& key, const std::string& value) { sections_[section][key] = value; }

    std::string getString(const std::string& section, const std::string& key) const {
        const auto s = sections_.find(section);
        if (s == sections_.end()) throw std::runtime_error("Section '" + section + "' not found");
        const auto k = s->second.find(key);
        if (k == s->second.end()) throw std::runtime_error("Key '" + key + "' not found in section '" + section + "'");
        return k->second;
    }
    int getInt(const std::string& section, const std::string& key) const {
        const std::string v = getString(section, key);
        try { return std::stoi(strip_us(v)); } catch (const std::exception&) { throw std::runtime_error("Cannot convert '" + v + "' to integer for key '" + key + "'"); }
    }
    int getInt(const std::string& s, const std::string& k, int dflt) const { try { return getInt(s, k); } catch (...) { return dflt; } }
    double getDouble(const std::string& section, const std::string& key) const {
        const std::string v = getString(section, key);
        try { return std::stod(strip_us(v)); } catch (const std::exception&) { throw std::runtime_error("Cannot convert '" + v + "' to double for key '" + key + "'"); }
    }
    double getDouble(const std::string& s, const std::string& k, double dflt) const { try { return getDouble(s, k); } catch (...) { return dflt; } }
    bool getBool(const std::string& section, const std::string& key) const {
        std::string v = getString(section, key);
        std::transform(v.begin(), v.end(), v.begin(), [](unsigned char c) { return std::tolower(c); });
        if (v == "true" || v == "1" || v == "yes" || v == "on") return true;
        if (v == "false" || v == "0" || v == "no" || v == "off") return false;
        throw std::runtime_error("Cannot convert '" + v + "' to boolean for key '" + key + "'");
    }
    bool getBool(const std::string& s, const std::string& k, bool dflt) const { try { return getBool(s, k); } catch (...) { return dflt; } }
    std::vector<double> getDoubleVector(const std::string& section, const std::string& key) const {
        std::vector<double> out; std::stringstream ss(getString(section, key)); std::string item;
        while (std::getline(ss, item, ',')) {
            trim(item); if (item.empty()) continue;
            try { out.push_back(std::stod(strip_us(item))); } catch (const std::exception&) { throw std::runtime_error("Cannot convert '" + item + "' to double in list for key '" + key + "'"); }
        }
        return out;
    }
    bool hasSection(const std::string& s) const { return sections_.count(s) != 0; }
    bool hasKey(const std::string& s, const std::string& k) const { const auto it = sections_.find(s); return it != sections_.end() && it->second.count(k) != 0; }
};

}  // namespace utility

// include/lattice.h:14-137
class Lattice {
    std::array<double, 2> a1_, a2_;
    std::vector<std::array<double, 2>> orbs_;
    int L1_, L2_, n_orb_;
    std::array<double, 2> b1_, b2_;
    std::vector<std::array<double, 2>> k_points_;
public:
    Lattice(const utility::parameters& params, const std::array<double, 2>& a1, const std::array<double, 2>& a2,
            const std::vector<std::array<double, 2>>& orbs)
        : a1_(a1), a2_(a2), orbs_(orbs), L1_(params.getInt("Lattice", "L1")), L2_(params.getInt("Lattice", "L2")), n_orb_((int)orbs.size()) {
        if (L1_ <= 0 || L2_ <= 0 || n_orb_ == 0) throw std::invalid_argument("Bad lattice dims");
        const double det = a1_[0] * a2_[1] - a1_[1] * a2_[0];
        if (std::abs(det) < 1e-12) throw std::invalid_argument("Singular lattice");
        b1_ = {2 * M_PI * a2_[1] / det / L1_, -2 * M_PI * a2_[0] / det / L1_};
        b2_ = {-2 * M_PI * a1_[1] / det / L2_, 2 * M_PI * a1_[0] / det / L2_};
        for (int nn = 0; nn < L1_; ++nn) for (int m = 0; m < L2_; ++m) {
            const int qx = nn - L1_ / 2 + 1, qy = m - L2_ / 2 + 1;
            k_points_.push_back({qx * b1_[0] + qy * b2_[0], qx * b1_[1] + qy * b2_[1]});
        }
    }
    int n_cells() const noexcept { return L1_ * L2_; }
    int n_sites() const noexcept { return L1_ * L2_ * n_orb_; }
    const int& L1() const noexcept { return L1_; }
    const int& L2() const noexcept { return L2_; }
    const int& n_orb() const noexcept { return n_orb_; }
    const std::array<double, 2>& a1() const noexcept { return a1_; }
    const std::array<double, 2>& a2() const noexcept { return a2_; }
    const std::vector<std::array<double, 2>>& k_points() const noexcept { return k_points_; }
    int site_neighbors(int idx, std::array<int, 2> delta, int orb) const {
        const int cell = idx / n_orb_, ux = cell % L1_, uy = cell / L1_;
        const int tx = ((ux + delta[0]) % L1_ + L1_) % L1_, ty = ((uy + delta[1]) % L2_ + L2_) % L2_;
        return (ty * L1_ + tx) * n_orb_ + orb;
    }
    void save_info(const std::string& filename) const {
        const std::string dir = filename.substr(0, filename.find_last_of("/\\"));
        if (!dir.empty() && dir != filename) { struct stat info; if (stat(dir.c_str(), &info) != 0) mkdir(dir.c_str(), 0755); }
        std::ofstream out(filename);
        if (out.is_open()) out << "L1 " << L1_ << "\nL2 " << L2_ << "\nn_orb " << n_orb_ << "\na1_x " << a1_[0] << "\na1_y " << a1_[1] << "\na2_x " << a2_[0] << "\na2_y " << a2_[1] << "\n";
    }
};

// include/field.h:13-84.  fields_ is nt x nv, column-major, 64-bit (arma::imat layout).
class GHQField {
    std::array<double, 4> gamma_{}, eta_{};
    static constexpr int proposal_[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
    int nt_ = 0, nv_ = 0;
    std::vector<int64_t> fields_;
public:
    GHQField() = default;
    GHQField(int nt, int nv, utility::random rng /* BY VALUE, as the reference: the caller's generator does not advance */) : nt_(nt), nv_(nv) {
        const double s6 = std::sqrt(6.0);
        gamma_ = {1.0 - s6 / 3.0, 1.0 + s6 / 3.0, 1.0 + s6 / 3.0, 1.0 - s6 / 3.0};
        eta_ = {-std::sqrt(2.0 * (3.0 + s6)), -std::sqrt(2.0 * (3.0 - s6)), std::sqrt(2.0 * (3.0 - s6)), std::sqrt(2.0 * (3.0 + s6))};
        fields_.resize((size_t)nt * nv);
        std::uniform_int_distribution<int> dist(0, 3);
        for (auto& f : fields_) f = dist(rng.get_generator());
    }
    double gamma(int f) const { return gamma_[f]; }
    double eta(int f) const { return eta_[f]; }
    const std::array<double, 4>& gamma_table() const { return gamma_; }
    const std::array<double, 4>& eta_table() const { return eta_; }
    int single_val(int l, int i) const { return (int)fields_[l + (size_t)nt_ * i]; }
    const std::vector<int64_t>& fields() const { return fields_; }
    std::vector<int64_t>& fields_mut() { return fields_; }
    int nv() const { return nv_; }
    int nt() const { return nt_; }
    void set_single_field(int l, int i, int v) { fields_[l + (size_t)nt_ * i] = v; }
    void set_fields(const std::vector<int64_t>& f) { fields_ = f; }
    static int proposal(int old_field, int k) { return proposal_[old_field][k]; }
    int propose_new_field(int old_field, utility::random rng /* BY VALUE (include/field.h:76) */) const {
        std::uniform_int_distribution<int> dist(0, 2);
        return proposal_[old_field][dist(rng.get_generator())];
    }
};

namespace dqmc_detail {
// exp(A) for a small-norm dense matrix: scaling and squaring with a degree-18 Taylor series
// (the reference calls arma::expmat, source/model.cpp:32-35; setup only).
inline std::vector<double> expm(const std::vector<double>& A, int n) {
    double nrm = 0.0;
    for (int j = 0; j < n; ++j) { double s = 0.0; for (int i = 0; i < n; ++i) s += std::fabs(A[i + (size_t)n * j]); nrm = std::max(nrm, s); }
    int sq = 0; while (std::ldexp(nrm, -sq) > 0.25) ++sq;
    const double sc = std::ldexp(1.0, -sq);
    std::vector<double> X(A.size()), term((size_t)n * n, 0.0), E((size_t)n * n, 0.0), T((size_t)n * n);
    for (size_t k = 0; k < A.size(); ++k) X[k] = A[k] * sc;
    for (int i = 0; i < n; ++i) { term[i + (size_t)n * i] = 1.0; E[i + (size_t)n * i] = 1.0; }
    auto mul = [n](const std::vector<double>& P, const std::vector<double>& Q, std::vector<double>& R) {
        std::fill(R.begin(), R.end(), 0.0);
        for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k) { const double q = Q[k + (size_t)n * j]; if (q == 0.0) continue; for (int i = 0; i < n; ++i) R[i + (size_t)n * j] += P[i + (size_t)n * k] * q; }
    };
    for (int k = 1; k <= 18; ++k) { mul(term, X, T); for (size_t e = 0; e < T.size(); ++e) { term[e] = T[e] / k; E[e] += term[e]; } }
    for (int s = 0; s < sq; ++s) { mul(E, E, T); E.swap(T); }
    return E;
}
inline void check(int rc) {
    if (rc == DQMC_OK) return;
    const std::string msg = dqmc_last_error();
    if (rc == DQMC_ERANGE) throw std::out_of_range(msg);       // LDRStack::operator[] (include/stackngf.h:61-66)
    if (rc == DQMC_EINVAL) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);                             // e.g. "QR decomposition failed in to_LDR" (source/stablelinalg.cpp:43-45)
}
}  // namespace dqmc_detail

// include/model.h:11-58
class AttractiveHubbard {
    double t_, mu_, g_, alpha_;
    int ns_, nt_;
    std::vector<double> expK_, invexpK_, expKhalf_, invexpKhalf_;
    GHQField fields_;
    utility::random& rng_;
    double dtau_ = 0.0; int L1_ = 0, L2_ = 0;
    std::vector<double> build_K_matrix(const Lattice& lat) const {           // source/model.cpp:39-60
        const int n = lat.n_sites(); std::vector<double> K((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) {
            K[i + (size_t)n * i] = -mu_;
            const int nx = lat.site_neighbors(i, {1, 0}, 0); K[i + (size_t)n * nx] = -t_; K[nx + (size_t)n * i] = -t_;
            const int ny = lat.site_neighbors(i, {0, 1}, 0); K[i + (size_t)n * ny] = -t_; K[ny + (size_t)n * i] = -t_;
        }
        return K;
    }
public:
    AttractiveHubbard(const utility::parameters& params, const Lattice& lat, utility::random& rng, double replica_beta) : rng_(rng) {
        t_ = params.getDouble("hubbard", "t"); mu_ = params.getDouble("hubbard", "mu");
        ns_ = lat.n_cells(); nt_ = (int)params.getDouble("simulation", "nt");
        const double U = params.getDouble("hubbard", "U"), dtau = replica_beta / nt_;
        dtau_ = dtau; L1_ = lat.L1(); L2_ = lat.L2();
        fields_ = GHQField(nt_, ns_, rng);
        g_ = std::sqrt(0.5 * std::abs(U) * dtau); alpha_ = -1.0;
        std::vector<double> K = build_K_matrix(lat), S(K.size());
        auto scaled = [&](double f) { for (size_t k = 0; k < K.size(); ++k) S[k] = f * K[k]; return dqmc_detail::expm(S, ns_); };
        expK_ = scaled(-dtau); invexpK_ = scaled(dtau); expKhalf_ = scaled(-0.5 * dtau); invexpKhalf_ = scaled(0.5 * dtau);
    }
    // Checkerboard break-up of exp(-dtau K) (the reference's README.md:40 lists it as future work; opt-in through
    // [simulation] checkerboard = true, see DQMC's constructor): the distinct bonds build_K_matrix sets to -t, split into groups
    // of disjoint pairs -- x bonds from even / odd columns and y bonds from even / odd rows when both lengths are even and >= 4,
    // a greedy edge colouring of the same bond list otherwise (dqmc_amd/model.py: checkerboard_groups is the same rule).
    std::vector<std::vector<std::array<int, 2>>> checkerboard_groups() const {
        if (L1_ < 2 || L2_ < 2) throw std::invalid_argument("checkerboard break-up needs L1, L2 >= 2");
        std::vector<std::vector<std::array<int, 2>>> groups;
        auto site = [&](int x, int y) { return ((y % L2_ + L2_) % L2_) * L1_ + ((x % L1_ + L1_) % L1_); };
        if (L1_ % 2 == 0 && L2_ % 2 == 0 && L1_ >= 4 && L2_ >= 4) {
            for (int p = 0; p < 2; ++p) { groups.emplace_back(); for (int y = 0; y < L2_; ++y) for (int x = p; x < L1_; x += 2) groups.back().push_back({site(x, y), site(x + 1, y)}); }
            for (int p = 0; p < 2; ++p) { groups.emplace_back(); for (int x = 0; x < L1_; ++x) for (int y = p; y < L2_; y += 2) groups.back().push_back({site(x, y), site(x, y + 1)}); }
            return groups;
        }
        std::vector<std::array<int, 2>> bonds;
        for (int i = 0; i < ns_; ++i) {
            const int x = i % L1_, y = i / L1_;
            for (int j : {site(x + 1, y), site(x, y + 1)}) {
                const std::array<int, 2> key{std::min(i, j), std::max(i, j)};
                bool seen = (i == j);
                for (const auto& b : bonds) if (std::min(b[0], b[1]) == key[0] && std::max(b[0], b[1]) == key[1]) { seen = true; break; }
                if (!seen) bonds.push_back({i, j});
            }
        }
        for (const auto& b : bonds) {
            bool placed = false;
            for (auto& g : groups) {
                bool free_ = true;
                for (const auto& q : g) if (q[0] == b[0] || q[1] == b[0] || q[0] == b[1] || q[1] == b[1]) { free_ = false; break; }
                if (free_) { g.push_back(b); placed = true; break; }
            }
            if (!placed) groups.push_back({b});
        }
        return groups;
    }
    double cosh_dtau_t() const { return std::cosh(dtau_ * t_); }
    double sinh_dtau_t() const { return std::sinh(dtau_ * t_); }
    double exp_dtau_mu() const { return std::exp(dtau_ * mu_); }
    const std::vector<double>& expK(int) const { return expK_; }
    const std::vector<double>& invexpK(int) const { return invexpK_; }
    const std::vector<double>& expK_half(int) const { return expKhalf_; }
    const std::vector<double>& invexpK_half(int) const { return invexpKhalf_; }
    std::vector<double> expV(int l, int) const { std::vector<double> v(ns_); for (int i = 0; i < ns_; ++i) v[i] = std::exp(g_ * fields_.eta(fields_.single_val(l, i))); return v; }
    std::vector<double> invexpV(int l, int) const { std::vector<double> v(ns_); for (int i = 0; i < ns_; ++i) v[i] = std::exp(-g_ * fields_.eta(fields_.single_val(l, i))); return v; }
    GHQField& fields() { return fields_; }
    const GHQField& fields() const { return fields_; }
    int nt() const { return nt_; }
    int ns() const { return ns_; }
    int n_flavor() const { return 1; }
    double g() const { return g_; }
    double alpha() const { return alpha_; }
    utility::random& rng() { return rng_; }
    // source/model.cpp:99-107
    std::pair<double, double> bosonic_ratio(int new_field, int old_field) const {
        const double d_eta = fields_.eta(new_field) - fields_.eta(old_field);
        const double br = std::exp(alpha_ * g_ * d_eta);
        return {br, (1.0 / br) - 1.0};
    }
};

// include/stackngf.h:15-29: the equal-time members; matrices are downloaded on demand.
struct GF {
    std::vector<double> Gtt0;     // host copy of Gtt[0] (column-major ns x ns), refreshed by DQMC::download
    double log_det_M = 0.0;
};

class DQMC;
// include/stackngf.h:34-77: a handle; the LDR triples live in HBM inside the engine.
class LDRStack {
    friend class DQMC;
    dqmc_engine* e_ = nullptr; size_t n_stack_ = 0; int n_ = 0;
public:
    LDRStack() = default;
    LDRStack(LDRStack&&) noexcept = default;
    LDRStack& operator=(LDRStack&&) noexcept = default;
    LDRStack(const LDRStack&) = delete;
    LDRStack& operator=(const LDRStack&) = delete;
    constexpr size_t size() const noexcept { return n_stack_; }
    struct LDR { std::vector<double> L, d, R; };
    LDR operator[](size_t idx) const {
        if (idx >= n_stack_) throw std::out_of_range("LDR Stack index out of bounds");
        LDR f; f.L.resize((size_t)n_ * n_); f.d.resize(n_); f.R.resize((size_t)n_ * n_);
        dqmc_detail::check(dqmc_get_stack(e_, (int)idx, f.L.data(), f.d.data(), f.R.data()));
        return f;
    }
};

namespace update {
// The random stream update::local_update consumes for ONE time slice
// (source/update.cpp:10-25), drawn with the reference's own calls in the reference's order:
// std::shuffle of the site order; then per site a proposal index from a BY-VALUE COPY of the
// generator (include/field.h:76-83 -- the shared generator does not advance) and the canonical
// uniform of bernoulli(p) (include/utility.h:34-37), which advances it by exactly two words
// whatever p is.  The copy is replayed from the two words the bernoulli draw is about to consume.
struct ReplayURNG {
    using result_type = std::mt19937::result_type;
    static constexpr result_type min() { return std::mt19937::min(); }
    static constexpr result_type max() { return std::mt19937::max(); }
    result_type w[2]; int k = 0; const std::mt19937* after;     // `after`: generator state once w[0], w[1] are consumed
    std::mt19937 tail; bool tail_init = false;
    result_type operator()() {
        if (k < 2) return w[k++];
        if (!tail_init) { tail = *after; tail_init = true; }    // Lemire rejection ran past two words (p ~ 2^-32): fall back to a real copy
        return tail();
    }
};
inline void draw_slice_stream(utility::random& rng, int nv, int32_t* perm, uint8_t* kprop, double* u) {
    std::vector<int> field_order(nv);
    for (int i = 0; i < nv; ++i) field_order[i] = i;
    std::shuffle(field_order.begin(), field_order.end(), rng.get_generator());      // source/update.cpp:14
    std::mt19937& gen = rng.get_generator();
    for (int idx = 0; idx < nv; ++idx) {
        perm[idx] = field_order[idx];
        ReplayURNG rep; rep.w[0] = gen(); rep.w[1] = gen(); rep.after = &gen;
        std::uniform_int_distribution<int> dist(0, 2);                             // include/field.h:79
        kprop[idx] = (uint8_t)dist(rep);
        // std::generate_canonical<double, 53>(mt19937): (w0 + w1 * 2^32) / 2^64, what bernoulli_distribution compares with p
        double c = (static_cast<double>(rep.w[0]) + static_cast<double>(rep.w[1]) * 4294967296.0) / 18446744073709551616.0;
        if (c >= 1.0) c = std::nextafter(1.0, 0.0);
        u[idx] = c;
    }
}
// source/update.cpp:34-45
inline int partner_rank(const int rank, const int world_size, const int exchange_attempt) {
    const bool even_attempt = (exchange_attempt % 2 == 0);
    const int off = even_attempt ? ((rank % 2 == 0) ? 1 : -1) : ((rank % 2 == 0) ? -1 : 1);
    return (rank + off + world_size) % world_size;
}
}  // namespace update

// include/dqmc.h:21-93
class DQMC {
    AttractiveHubbard& model_;
    dqmc_engine* e_ = nullptr;
    int nt_, n_stab_, n_stack_;
    bool isUnequalTime_ = false;
    std::vector<int32_t> perm_; std::vector<uint8_t> kprop_; std::vector<double> u_;
    void draw_half_sweep(bool forward) {
        const int ns = model_.ns();
        for (int step = 0; step < nt_; ++step) {
            const int l = forward ? step : nt_ - 1 - step;       // slices in the order the sweep visits them
            update::draw_slice_stream(model_.rng(), ns, perm_.data() + (size_t)l * ns, kprop_.data() + (size_t)l * ns, u_.data() + (size_t)l * ns);
        }
    }
public:
    DQMC(const utility::parameters& params, AttractiveHubbard& model, int device = 0) : model_(model) {
        nt_ = params.getInt("simulation", "nt"); n_stab_ = params.getInt("simulation", "n_stab");
        n_stack_ = (int)std::ceil(static_cast<double>(nt_) / n_stab_);
        try { isUnequalTime_ = params.getBool("simulation", "isMeasureUnequalTime"); } catch (...) { isUnequalTime_ = false; }   // source/dqmc.cpp:9
        const GHQField& f = model.fields();
        dqmc_detail::check(dqmc_create(&e_, device, model.ns(), nt_, n_stab_, model.g(), f.gamma_table().data(), f.eta_table().data(),
                                       model.expK(0).data(), model.invexpK(0).data()));
        if (params.getBool("simulation", "checkerboard", false)) {      // extension key (README.md:40 of the reference: future work); default: dense exp
            const auto groups = model.checkerboard_groups();
            std::vector<int32_t> bonds, sizes;
            for (const auto& g : groups) { sizes.push_back((int32_t)g.size()); for (const auto& b : g) { bonds.push_back(b[0]); bonds.push_back(b[1]); } }
            const double c = model.cosh_dtau_t(), s = model.sinh_dtau_t(), f = model.exp_dtau_mu();
            dqmc_detail::check(dqmc_set_checkerboard(e_, (int)groups.size(), bonds.data(), sizes.data(), &c, &s, &f));
        }
        const size_t cnt = (size_t)nt_ * model.ns(); perm_.resize(cnt); kprop_.resize(cnt); u_.resize(cnt);
    }
    ~DQMC() { dqmc_destroy(e_); }
    DQMC(const DQMC&) = delete;
    DQMC& operator=(const DQMC&) = delete;
    dqmc_engine* handle() { return e_; }
    double acc_rate() { dqmc_stats s; dqmc_detail::check(dqmc_get_stats(e_, &s)); return s.acc_rate; }
    double max_err() { dqmc_stats s; dqmc_detail::check(dqmc_get_stats(e_, &s)); return s.max_err; }
    double mean_err() { dqmc_stats s; dqmc_detail::check(dqmc_get_stats(e_, &s)); return s.sum_err / s.n_err; }
    // source/dqmc.cpp:43-59: uploads the model's fields and rebuilds every stack[i] = B(beta, tau_i)
    LDRStack init_stacks(int /*flv*/) {
        dqmc_detail::check(dqmc_set_fields(e_, model_.fields().fields().data()));
        dqmc_detail::check(dqmc_init(e_));
        LDRStack s; s.e_ = e_; s.n_stack_ = (size_t)n_stack_; s.n_ = model_.ns(); return s;
    }
    // a handle onto the engine's current stack without rebuilding it
    LDRStack stack_handle() { LDRStack s; s.e_ = e_; s.n_stack_ = (size_t)n_stack_; s.n_ = model_.ns(); return s; }
    // source/dqmc.cpp:61-72
    GF init_greenfunctions(LDRStack&) { GF g; download(g); return g; }
    void download(GF& g) {
        g.Gtt0.resize((size_t)model_.ns() * model_.ns());
        dqmc_detail::check(dqmc_get_G(e_, g.Gtt0.data()));
        dqmc_detail::check(dqmc_get_logdet(e_, &g.log_det_M));
    }
    // pulls the device's HS fields back into model.fields() (they change on device during sweeps)
    void download_fields() { dqmc_detail::check(dqmc_get_fields(e_, model_.fields().fields_mut().data())); }
    // source/dqmc.cpp:337-396 / :398-456; asynchronous, greens are refreshed by download()
    void sweep_0_to_beta(std::vector<GF>&, std::vector<LDRStack>&) { draw_half_sweep(true); dqmc_detail::check(dqmc_sweep_0_to_beta(e_, perm_.data(), kprop_.data(), u_.data())); }
    void sweep_beta_to_0(std::vector<GF>&, std::vector<LDRStack>&) { draw_half_sweep(false); dqmc_detail::check(dqmc_sweep_beta_to_0(e_, perm_.data(), kprop_.data(), u_.data())); }
    // source/dqmc.cpp:458-515: a no-op unless [simulation] isMeasureUnequalTime = true (:461-463); the Gtt / Gt0 / G0t series stay in HBM,
    // download_tau() fetches one slice (GF::Gtt[l], Gt0[l], G0t[l] of include/stackngf.h:15-29)
    void sweep_unequalTime(std::vector<GF>&, std::vector<LDRStack>&) { if (isUnequalTime_) dqmc_detail::check(dqmc_sweep_unequal_time(e_)); }
    enum Tau { Gtt = 0, Gt0 = 1, G0t = 2 };
    std::vector<double> download_tau(Tau which, int l) {
        std::vector<double> m((size_t)model_.ns() * model_.ns());
        dqmc_detail::check(dqmc_get_G_tau(e_, (int)which, l, m.data())); return m;
    }
    bool isUnequalTime() const { return isUnequalTime_; }
    // source/dqmc.cpp:288-315 (called at source/main.cpp:161-163 when [simulation] symmetric = true): GF_tosymm.Gtt[0] = invexpK_half *
    // Gtt[0] * expK_half.  GF_asymm is the engine's device-resident G, so the second argument only keeps the reference's signature.  The
    // reference also half-warps every slice of the unequal-time series (:306-313); here those stay in HBM and half_warp_tau() returns one
    // half-warped slice on demand, like download_tau().
    void half_warp(std::vector<GF>& GF_tosymm, std::vector<GF>& /*GF_asymm*/) {
        const size_t nn = (size_t)model_.ns() * model_.ns();
        for (int flv = 0; flv < model_.n_flavor(); ++flv) {
            GF_tosymm.at(flv).Gtt0.resize(nn);
            dqmc_detail::check(dqmc_half_warp(e_, model_.expK_half(flv).data(), model_.invexpK_half(flv).data(), -1, 0, GF_tosymm[flv].Gtt0.data()));
        }
    }
    std::vector<double> half_warp_tau(Tau which, int l) {
        std::vector<double> m((size_t)model_.ns() * model_.ns());
        dqmc_detail::check(dqmc_half_warp(e_, model_.expK_half(0).data(), model_.invexpK_half(0).data(), (int)which, l, m.data())); return m;
    }
    // greenTau / doublonTau / currxxTau (source/model.cpp:290-394) of the last sweep_unequalTime in displacement space:
    // [3][nt + 1][L1*L2]; measure_unequal() adds them to the device-side bin, fetch_unequal_bin() returns the bin averages
    std::vector<double> measure_unequal_time(const Lattice& lat) {
        std::vector<double> out((size_t)3 * (nt_ + 1) * lat.L1() * lat.L2());
        dqmc_detail::check(dqmc_measure_unequal_time(e_, lat.L1(), lat.L2(), 0, out.data())); return out;
    }
    void measure_unequal(const Lattice& lat) { if (isUnequalTime_) dqmc_detail::check(dqmc_measure_unequal_time(e_, lat.L1(), lat.L2(), 1, nullptr)); }
    std::vector<double> fetch_unequal_bin(const Lattice& lat, long long* n_out = nullptr) {
        std::vector<double> out((size_t)3 * (nt_ + 1) * lat.L1() * lat.L2()); int64_t cnt = 0;
        dqmc_detail::check(dqmc_measure_unequal_fetch(e_, out.data(), &cnt, 1));
        if (cnt > 0) for (double& x : out) x /= (double)cnt;
        if (n_out) *n_out = cnt;
        return out;
    }
    double global_action() { double S; dqmc_detail::check(dqmc_global_action(e_, &S)); return S; }   // source/model.cpp:140-159
    // ---- equal-time observables on the device (source/model.cpp:167-288, include/measurementh5.h:13-66) ----
    struct EqualTime { double density = 0, doubleOcc = 0, swave = 0; std::vector<double> densityCorr_r; long long n = 0; };
    // one evaluation on the current Gtt[0] (what measurements.measure(greens, lat) computes, source/main.cpp:165)
    EqualTime measure_equal_time(const Lattice& lat) {
        EqualTime m; double sc[3]; m.densityCorr_r.resize((size_t)lat.L1() * lat.L2());
        dqmc_detail::check(dqmc_measure_equal_time(e_, lat.L1(), lat.L2(), sc, m.densityCorr_r.data()));
        m.density = sc[0]; m.doubleOcc = sc[1]; m.swave = sc[2]; m.n = 1; return m;
    }
    // bin accumulation: measure() per sweep stays on the device (asynchronous), fetch_bin() returns the bin AVERAGES and starts a new bin
    void measure(const Lattice& lat) { dqmc_detail::check(dqmc_measure_accumulate(e_, lat.L1(), lat.L2())); }
    EqualTime fetch_bin(const Lattice& lat) {
        EqualTime m; double sc[3]; int64_t cnt = 0; m.densityCorr_r.resize((size_t)lat.L1() * lat.L2());
        dqmc_detail::check(dqmc_measure_fetch(e_, sc, m.densityCorr_r.data(), &cnt, 1));
        const double inv = cnt > 0 ? 1.0 / (double)cnt : 0.0;
        m.density = sc[0] * inv; m.doubleOcc = sc[1] * inv; m.swave = sc[2] * inv; m.n = cnt;
        for (double& x : m.densityCorr_r) x *= inv;
        return m;
    }
};

namespace update {
// The canonical uniform that rng.bernoulli(p) compares with p (include/utility.h:34-37): std::bernoulli_distribution on
// std::mt19937 draws std::generate_canonical<double, 53>, i.e. exactly two words whatever p is, so the decision
// "bernoulli(p)" can be taken wherever p becomes known as "u < p" (checked against the real thing in tests/test_host.py).
inline double draw_bernoulli_uniform(utility::random& rng) {
    std::mt19937& gen = rng.get_generator();
    const std::mt19937::result_type w0 = gen(), w1 = gen();
    double c = (static_cast<double>(w0) + static_cast<double>(w1) * 4294967296.0) / 18446744073709551616.0;
    if (c >= 1.0) c = std::nextafter(1.0, 0.0);
    return c;
}

// source/update.cpp:47-117.  `comm` plays MPI_COMM_WORLD (one rank per GPU): the field swap, the trial
// re-initialisation, the two action exchanges, the decision by the lower rank and the restoring re-initialisation run
// inside dqmc_replica_exchange_round on HBM-resident state (RCCL: device to device over xGMI); the host side keeps
// what the reference keeps on the host -- the attempt counter, the partner rule, the decider's bernoulli draw from the
// rank's utility::random, rank 0's acceptance counter -- and refreshes model.fields() and the GF / LDRStack handles.
inline dqmc_exchange_result replica_exchange(int rank, int world_size, utility::random& rng, int& exchange_attempt, int& exchange_accepted,
                                             AttractiveHubbard& model, DQMC& sim, std::vector<GF>& greens, std::vector<LDRStack>& stacks,
                                             dqmc_comm* comm) {
    exchange_attempt++;
    dqmc_exchange_result res{}; res.partner = -1;
    const int partner = partner_rank(rank, world_size, exchange_attempt);
    if (partner < 0 || partner >= world_size) return res;                    // partner out of bounds: do nothing (:55-57)
    if (dqmc_comm_rank(comm) != rank || dqmc_comm_world_size(comm) != world_size) throw std::invalid_argument("replica_exchange: communicator does not match rank / world_size");
    const double u = rank < partner ? draw_bernoulli_uniform(rng) : 0.0;      // only the decider's generator advances (:96)
    dqmc_detail::check(dqmc_replica_exchange_round(sim.handle(), comm, exchange_attempt, u, &res));
    if (rank == 0) exchange_accepted += res.accepted;                        // :99-101
    sim.download_fields();                                                   // model.fields(): partner's on acceptance, restored otherwise
    for (int flv = 0; flv < model.n_flavor(); ++flv) { stacks[flv] = sim.stack_handle(); greens[flv] = sim.init_greenfunctions(stacks[flv]); }
    return res;
}

// In-process stand-in for MPI_COMM_WORLD: `world` replicas run as threads of one process (one engine each, on one or
// several GPUs) and meet in a mailbox.  endpoint(rank) + InProcessHub::sendrecv are the (user, fn) pair of
// dqmc_comm_create_callbacks.  Used by the driver's single-process parallel-tempering mode and by the tests.
class InProcessHub {
    struct Key { int src, dst, tag; bool operator<(const Key& o) const { return std::tie(src, dst, tag) < std::tie(o.src, o.dst, o.tag); } };
    std::mutex mu_; std::condition_variable cv_;
    std::map<Key, std::deque<std::vector<char>>> box_;
    int world_; bool aborted_ = false;
public:
    struct Endpoint { InProcessHub* hub; int rank; };
private:
    std::vector<Endpoint> eps_;
public:
    explicit InProcessHub(int world) : world_(world) { for (int r = 0; r < world; ++r) eps_.push_back({this, r}); }
    int world() const { return world_; }
    void abort() { std::lock_guard<std::mutex> lk(mu_); aborted_ = true; cv_.notify_all(); }
    void* endpoint(int rank) { return &eps_.at(rank); }
    static int sendrecv(void* user, const void* send, void* recv, size_t bytes, int partner, int tag) {
        Endpoint* ep = static_cast<Endpoint*>(user); InProcessHub& h = *ep->hub;
        if (partner < 0 || partner >= h.world_ || partner == ep->rank) return -1;
        std::unique_lock<std::mutex> lk(h.mu_);
        const char* p = static_cast<const char*>(send);
        h.box_[Key{ep->rank, partner, tag}].emplace_back(p, p + bytes);
        h.cv_.notify_all();
        auto& q = h.box_[Key{partner, ep->rank, tag}];
        h.cv_.wait(lk, [&] { return !q.empty() || h.aborted_; });
        if (q.empty()) return -3;                                          // a replica failed: everybody leaves
        if (q.front().size() != bytes) return -2;
        std::memcpy(recv, q.front().data(), bytes); q.pop_front();
        return 0;
    }
    dqmc_comm* make_comm(int rank) {
        dqmc_comm* c = nullptr;
        dqmc_detail::check(dqmc_comm_create_callbacks(&c, world_, rank, &InProcessHub::sendrecv, endpoint(rank)));
        return c;
    }
};
}  // namespace update
