// results_h5.hpp -- the on-disk results of a run in the reference's layout, so that the
// reference's scripts/analysis.py keeps working on this engine's output:
//
//   results/data_<rank>.h5                                        MeasurementManager::saveToHDF5, include/measurementh5.h:277-362
//     /bin_<k>/scalar/{density,doubleOcc,swave}                   1-element fp64 datasets            (include/h5utils.h:9-26)
//     /bin_<k>/equaltime/densityCorr                              (L1, L2, 1)       chi(dx, dy)       (h5utils.h:50-81: the cube is
//     /bin_<k>/unequaltime/{greenTau,doublonTau,currxxTau}        (L1, L2, nt + 1)  chi(dx, dy, tau)   written so that a C-order reader sees [i][j][k])
//     /binK_<k>/equaltime/..., /binK_<k>/unequaltime/...          (L1, L2, n_tau, 2) the lattice Fourier transform, re / im last (h5utils.h:83-119)
//
// The displacement-space data (transform::chi_site_to_chi_r, measurementh5.h:13-66) comes straight from the device bins
// (dqmc_measure_fetch / dqmc_measure_unequal_fetch: index dx_idx + L1 * dy_idx, dx_idx = dx + L1/2 - 1); the k-space transform
// (transform::chi_r_to_chi_k, measurementh5.h:77-117) runs here on the host, once per bin.
//
// libhdf5 is resolved at run time with dlopen (the image ships HDF5 1.10.6 under /opt/conda/lib; DQMC_HDF5_LIB overrides the
// path), so neither the product library nor the driver has a link-time dependency on it; without the library the
// writer throws std::runtime_error on open.  Known deviation, on purpose: the reference maps k-point number kidx to
// (kidx / L1, kidx % L2) (measurementh5.h:99-100), which is only a bijection for L1 == L2; this writer uses
// (kidx / L2, kidx % L2), identical for square lattices.
#pragma once

#include <array>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <dlfcn.h>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace transform {
// chi_r: [n_tau][L1 * L2] with the displacement bin dx_idx + L1 * dy_idx fastest.  Returns chi_k[(kx * L2 + ky) * n_tau + t].
inline std::vector<std::complex<double>> chi_r_to_chi_k(const std::vector<double>& chi_r, int n_tau, int L1, int L2,
                                                         const std::array<double, 2>& a1, const std::array<double, 2>& a2,
                                                         const std::vector<std::array<double, 2>>& kpts) {
    std::vector<std::complex<double>> chi_k((size_t)L1 * L2 * n_tau, std::complex<double>(0.0, 0.0));
    const int nk = (int)kpts.size();
    for (int kidx = 0; kidx < nk; ++kidx) {
        const auto& k = kpts[kidx];
        const int kx_idx = kidx / L2, ky_idx = kidx % L2;
        for (int x_idx = 0; x_idx < L1; ++x_idx)
            for (int y_idx = 0; y_idx < L2; ++y_idx) {
                const double dx = (x_idx - (L1 / 2 - 1)) * a1[0] + (y_idx - (L2 / 2 - 1)) * a2[0];
                const double dy = (x_idx - (L1 / 2 - 1)) * a1[1] + (y_idx - (L2 / 2 - 1)) * a2[1];
                const double phase = k[0] * dx + k[1] * dy;
                const std::complex<double> w(std::cos(phase), -std::sin(phase));
                for (int t = 0; t < n_tau; ++t)
                    chi_k[((size_t)kx_idx * L2 + ky_idx) * n_tau + t] += chi_r[(size_t)t * L1 * L2 + x_idx + (size_t)L1 * y_idx] * w;
            }
    }
    return chi_k;
}
}  // namespace transform

namespace hdf5 {
using hid_t = int64_t;              // HDF5 >= 1.10
using hsize_t = unsigned long long;
using herr_t = int;

// the dozen libhdf5 entry points the writer and the read-back helper need
struct Api {
    void* lib = nullptr;
    herr_t (*H5open)() = nullptr;
    hid_t (*H5Fcreate)(const char*, unsigned, hid_t, hid_t) = nullptr;
    hid_t (*H5Fopen)(const char*, unsigned, hid_t) = nullptr;
    herr_t (*H5Fclose)(hid_t) = nullptr;
    herr_t (*H5Fflush)(hid_t, int) = nullptr;
    hid_t (*H5Gcreate2)(hid_t, const char*, hid_t, hid_t, hid_t) = nullptr;
    herr_t (*H5Gclose)(hid_t) = nullptr;
    hid_t (*H5Screate_simple)(int, const hsize_t*, const hsize_t*) = nullptr;
    herr_t (*H5Sclose)(hid_t) = nullptr;
    hid_t (*H5Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
    hid_t (*H5Dopen2)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*) = nullptr;
    herr_t (*H5Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*) = nullptr;
    hid_t (*H5Dget_space)(hid_t) = nullptr;
    int (*H5Sget_simple_extent_ndims)(hid_t) = nullptr;
    int (*H5Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*) = nullptr;
    herr_t (*H5Dclose)(hid_t) = nullptr;
    herr_t (*H5Eset_auto2)(hid_t, void*, void*) = nullptr;
    hid_t native_double = -1;
    std::string path;

    static Api& get() {
        static Api api;
        if (api.lib) return api;
        std::vector<std::string> cand;
        if (const char* e = std::getenv("DQMC_HDF5_LIB")) cand.push_back(e);
        for (const char* c : {"/opt/conda/lib/libhdf5.so", "libhdf5.so.103", "libhdf5.so", "libhdf5_serial.so"}) cand.push_back(c);
        for (const auto& c : cand) { api.lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL); if (api.lib) { api.path = c; break; } }
        if (!api.lib) throw std::runtime_error("results_h5: libhdf5 not found (set DQMC_HDF5_LIB)");
        auto sym = [&](const char* name) { void* p = dlsym(api.lib, name); if (!p) throw std::runtime_error(std::string("results_h5: missing symbol ") + name); return p; };
#define DQ_H5SYM(f) api.f = reinterpret_cast<decltype(api.f)>(sym(#f))
        DQ_H5SYM(H5open); DQ_H5SYM(H5Fcreate); DQ_H5SYM(H5Fopen); DQ_H5SYM(H5Fclose); DQ_H5SYM(H5Fflush); DQ_H5SYM(H5Gcreate2); DQ_H5SYM(H5Gclose);
        DQ_H5SYM(H5Screate_simple); DQ_H5SYM(H5Sclose); DQ_H5SYM(H5Dcreate2); DQ_H5SYM(H5Dopen2); DQ_H5SYM(H5Dwrite); DQ_H5SYM(H5Dread);
        DQ_H5SYM(H5Dget_space); DQ_H5SYM(H5Sget_simple_extent_ndims); DQ_H5SYM(H5Sget_simple_extent_dims); DQ_H5SYM(H5Dclose); DQ_H5SYM(H5Eset_auto2);
#undef DQ_H5SYM
        if (api.H5open() < 0) throw std::runtime_error("results_h5: H5open failed");
        api.H5Eset_auto2(0 /* H5E_DEFAULT */, nullptr, nullptr);      // failures are reported through exceptions, not HDF5's stderr stack dumps
        api.native_double = *reinterpret_cast<hid_t*>(sym("H5T_NATIVE_DOUBLE_g"));   // what the H5T_NATIVE_DOUBLE macro expands to
        return api;
    }
};
constexpr unsigned ACC_RDONLY = 0u, ACC_TRUNC = 2u;      // H5F_ACC_RDONLY, H5F_ACC_TRUNC
constexpr hid_t P_DEFAULT = 0, S_ALL = 0;                // H5P_DEFAULT, H5S_ALL

// C-order dataset of doubles: dims as the reference declares them, data[i][j][k]...
inline void write_array(hid_t loc, const std::string& name, const std::vector<hsize_t>& dims, const double* data) {
    Api& h = Api::get();
    const hid_t sp = h.H5Screate_simple((int)dims.size(), dims.data(), nullptr);
    const hid_t ds = h.H5Dcreate2(loc, name.c_str(), h.native_double, sp, P_DEFAULT, P_DEFAULT, P_DEFAULT);
    const herr_t st = ds < 0 ? -1 : h.H5Dwrite(ds, h.native_double, S_ALL, S_ALL, P_DEFAULT, data);
    if (ds >= 0) h.H5Dclose(ds);
    h.H5Sclose(sp);
    if (st < 0) throw std::runtime_error("Failed to write " + name + " to HDF5 file");
}
// read-back helper (tests, tooling): dims and C-order data of a fp64 dataset
inline std::vector<double> read_array(const std::string& file, const std::string& dataset, std::vector<hsize_t>* dims_out) {
    Api& h = Api::get();
    const hid_t f = h.H5Fopen(file.c_str(), ACC_RDONLY, P_DEFAULT);
    if (f < 0) throw std::runtime_error("Failed to open HDF5 file: " + file);
    const hid_t ds = h.H5Dopen2(f, dataset.c_str(), P_DEFAULT);
    if (ds < 0) { h.H5Fclose(f); throw std::runtime_error("No dataset " + dataset + " in " + file); }
    const hid_t sp = h.H5Dget_space(ds);
    const int nd = h.H5Sget_simple_extent_ndims(sp);
    std::vector<hsize_t> dims((size_t)(nd > 0 ? nd : 0));
    if (nd > 0) h.H5Sget_simple_extent_dims(sp, dims.data(), nullptr);
    size_t cnt = 1; for (hsize_t d : dims) cnt *= (size_t)d;
    std::vector<double> out(cnt);
    const herr_t st = h.H5Dread(ds, h.native_double, S_ALL, S_ALL, P_DEFAULT, out.data());
    h.H5Sclose(sp); h.H5Dclose(ds); h.H5Fclose(f);
    if (st < 0) throw std::runtime_error("Failed to read " + dataset);
    if (dims_out) *dims_out = dims;
    return out;
}
}  // namespace hdf5

// One bin of measurements, already averaged (what MeasurementManager holds when saveToHDF5 runs, measurementh5.h:232-250)
struct ResultsBin {
    double density = 0.0, doubleOcc = 0.0, swave = 0.0;
    std::vector<double> densityCorr_r;                 // [L1 * L2]
    std::vector<double> unequal_r;                     // empty, or [3][nt + 1][L1 * L2]: greenTau, doublonTau, currxxTau
    int n_tau = 0;                                     // nt + 1 when unequal_r is present
};

class ResultsWriter {
    hdf5::hid_t file_ = -1;
    int current_bin_ = 0;
    int L1_, L2_;
    std::array<double, 2> a1_, a2_;
    std::vector<std::array<double, 2>> kpts_;

    // [n_tau][L1*L2] (bin fastest) -> C-order [dx][dy][tau]
    std::vector<double> to_cube(const double* r, int n_tau) const {
        std::vector<double> c((size_t)L1_ * L2_ * n_tau);
        for (int t = 0; t < n_tau; ++t) for (int y = 0; y < L2_; ++y) for (int x = 0; x < L1_; ++x)
            c[((size_t)x * L2_ + y) * n_tau + t] = r[(size_t)t * L1_ * L2_ + x + (size_t)L1_ * y];
        return c;
    }
    void write_pair(hdf5::hid_t g_r, hdf5::hid_t g_k, const std::string& name, const double* r, int n_tau) const {
        using hdf5::hsize_t;
        const std::vector<double> cube = to_cube(r, n_tau);
        hdf5::write_array(g_r, name, {(hsize_t)L1_, (hsize_t)L2_, (hsize_t)n_tau}, cube.data());
        const std::vector<double> rr(r, r + (size_t)n_tau * L1_ * L2_);
        const auto chi_k = transform::chi_r_to_chi_k(rr, n_tau, L1_, L2_, a1_, a2_, kpts_);
        hdf5::write_array(g_k, name, {(hsize_t)L1_, (hsize_t)L2_, (hsize_t)n_tau, 2}, reinterpret_cast<const double*>(chi_k.data()));
    }
public:
    // directory/data_<rank>.h5, truncated like the reference's H5Fcreate(..., H5F_ACC_TRUNC) (include/h5utils.h:121-127)
    ResultsWriter(const std::string& directory, int rank, int L1, int L2, const std::array<double, 2>& a1, const std::array<double, 2>& a2,
                  const std::vector<std::array<double, 2>>& k_points)
        : L1_(L1), L2_(L2), a1_(a1), a2_(a2), kpts_(k_points) {
        hdf5::Api& h = hdf5::Api::get();
        struct stat info;
        if (stat(directory.c_str(), &info) != 0) mkdir(directory.c_str(), 0755);
        const std::string filename = directory + "/data_" + std::to_string(rank) + ".h5";
        file_ = h.H5Fcreate(filename.c_str(), hdf5::ACC_TRUNC, hdf5::P_DEFAULT, hdf5::P_DEFAULT);
        if (file_ < 0) throw std::runtime_error("Failed to create HDF5 file: " + filename);
    }
    ResultsWriter(const ResultsWriter&) = delete;
    ResultsWriter& operator=(const ResultsWriter&) = delete;
    ~ResultsWriter() { if (file_ >= 0) hdf5::Api::get().H5Fclose(file_); }
    int bins_written() const { return current_bin_; }

    void write_bin(const ResultsBin& b) {
        hdf5::Api& h = hdf5::Api::get();
        if ((int)b.densityCorr_r.size() != L1_ * L2_) throw std::invalid_argument("ResultsWriter: densityCorr_r must hold L1*L2 values");
        if (!b.unequal_r.empty() && (b.n_tau <= 0 || b.unequal_r.size() != (size_t)3 * b.n_tau * L1_ * L2_))
            throw std::invalid_argument("ResultsWriter: unequal_r must hold 3*(nt+1)*L1*L2 values");
        const std::string gr = "/bin_" + std::to_string(current_bin_), gk = "/binK_" + std::to_string(current_bin_);
        auto group = [&](hdf5::hid_t loc, const char* name) {
            const hdf5::hid_t g = h.H5Gcreate2(loc, name, hdf5::P_DEFAULT, hdf5::P_DEFAULT, hdf5::P_DEFAULT);
            if (g < 0) throw std::runtime_error(std::string("Failed to create HDF5 group ") + name);
            return g;
        };
        const hdf5::hid_t g_r = group(file_, gr.c_str()), g_k = group(file_, gk.c_str());
        const hdf5::hid_t sc_r = group(g_r, "scalar"), eq_r = group(g_r, "equaltime"), ut_r = group(g_r, "unequaltime");
        const hdf5::hid_t eq_k = group(g_k, "equaltime"), ut_k = group(g_k, "unequaltime");
        hdf5::write_array(sc_r, "density", {1}, &b.density);
        hdf5::write_array(sc_r, "doubleOcc", {1}, &b.doubleOcc);
        hdf5::write_array(sc_r, "swave", {1}, &b.swave);
        write_pair(eq_r, eq_k, "densityCorr", b.densityCorr_r.data(), 1);
        if (!b.unequal_r.empty()) {
            static const char* names[3] = {"greenTau", "doublonTau", "currxxTau"};
            for (int o = 0; o < 3; ++o) write_pair(ut_r, ut_k, names[o], b.unequal_r.data() + (size_t)o * b.n_tau * L1_ * L2_, b.n_tau);
        }
        for (hdf5::hid_t g : {sc_r, eq_r, ut_r, eq_k, ut_k, g_r, g_k}) h.H5Gclose(g);
        h.H5Fflush(file_, 1 /* H5F_SCOPE_GLOBAL */);
        ++current_bin_;
    }
};
