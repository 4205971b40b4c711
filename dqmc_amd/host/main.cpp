// main.cpp -- a driver with the loop shape of the reference's source/main.cpp:14-214 on top of the
// host facade: read ./parameters.in (or argv[1]), build Lattice / AttractiveHubbard / DQMC,
// thermalise, sweep, report time, acceptance and wrap error.  One process = one Markov chain on one
// GPU (the reference's MPI rank); the multi-GPU launcher and replica exchange over RCCL are in
// bench.py / dqmc_amd/replica.py.  Observables are accumulated per bin on the device, printed, and
// written to results/data_<rank>.h5 in the reference's layout (results_h5.hpp; argv[4] = rank, default 0;
// DQMC_NO_HDF5=1 or a missing libhdf5 turns the file off with a warning, the run itself does not depend on it).
#include "dqmc_host.hpp"
#include "results_h5.hpp"

#include <chrono>
#include <cstdio>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <memory>

int main(int argc, char** argv) {
    try {
        const std::string pfile = argc > 1 ? argv[1] : "parameters.in";
        const int device = argc > 2 ? std::atoi(argv[2]) : 0;
        const unsigned seed = argc > 3 ? (unsigned)std::strtoul(argv[3], nullptr, 10) : (unsigned)std::time(nullptr);   // source/main.cpp:37
        const int rank = argc > 4 ? std::atoi(argv[4]) : 0;
        utility::parameters params(pfile);
        utility::random rng(seed);
        const double my_beta = params.getDouble("simulation", "beta");
        const int n_sweeps = params.getInt("simulation", "n_sweeps"), n_therms = params.getInt("simulation", "n_therms"), n_bins = params.getInt("simulation", "n_bins");

        Lattice lat(params, {1.0, 0.0}, {0.0, 1.0}, {{0.0, 0.0}});
        lat.save_info("results/info");
        AttractiveHubbard model(params, lat, rng, my_beta);
        const int n_flavor = model.n_flavor();
        DQMC sim(params, model, device);

        std::vector<LDRStack> propagation_stacks(n_flavor);
        std::vector<GF> greens(n_flavor);
        for (int flv = 0; flv < n_flavor; flv++) {
            propagation_stacks[flv] = sim.init_stacks(flv);
            greens[flv] = sim.init_greenfunctions(propagation_stacks[flv]);
        }
        std::cout << "Standard DQMC run (Parallel Tempering disabled), backend " << dqmc_backend() << ", log det M = "
                  << std::setprecision(12) << greens[0].log_det_M << "\n";

        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n_therms; ++i) { sim.sweep_0_to_beta(greens, propagation_stacks); sim.sweep_beta_to_0(greens, propagation_stacks); }
        sim.download(greens[0]);
        const double dt_therm = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "Thermalization done in " << dt_therm << " seconds\n";

        std::unique_ptr<ResultsWriter> results;
        if (!std::getenv("DQMC_NO_HDF5")) {
            try { results.reset(new ResultsWriter("results", rank, lat.L1(), lat.L2(), lat.a1(), lat.a2(), lat.k_points())); }
            catch (const std::exception& e) { std::cerr << "warning: no HDF5 output (" << e.what() << ")\n"; }
        }
        const long bin_sweeps = (long)n_bins * n_sweeps;
        const auto t1 = std::chrono::steady_clock::now();
        for (long isweep = 1; isweep <= bin_sweeps; ++isweep) {
            sim.sweep_0_to_beta(greens, propagation_stacks);
            sim.sweep_beta_to_0(greens, propagation_stacks);
            sim.sweep_unequalTime(greens, propagation_stacks);
            sim.measure(lat);                                          // measurements.measure(greens, lat), source/main.cpp:165 -- on the device, asynchronous
            sim.measure_unequal(lat);                                  // the dynamical observables of the same call, when isMeasureUnequalTime
            if (isweep % n_sweeps == 0) {                              // measurements.accumulate(lat), :167-169: one bin done
                const DQMC::EqualTime bin = sim.fetch_bin(lat);
                std::cout << "bin " << isweep / n_sweeps << " (" << bin.n << " sweeps): density " << std::setprecision(8) << bin.density
                          << "  doubleOcc " << bin.doubleOcc << "  swave " << bin.swave << "  densityCorr(r=0) "
                          << bin.densityCorr_r[(size_t)(lat.L1() / 2 - 1) + (size_t)lat.L1() * (lat.L2() / 2 - 1)] << '\n';
                ResultsBin rb; rb.density = bin.density; rb.doubleOcc = bin.doubleOcc; rb.swave = bin.swave; rb.densityCorr_r = bin.densityCorr_r;
                if (sim.isUnequalTime()) {
                    long long nu = 0; const std::vector<double> ut = sim.fetch_unequal_bin(lat, &nu);
                    rb.unequal_r = ut; rb.n_tau = params.getInt("simulation", "nt") + 1;
                    const size_t r0 = (size_t)(lat.L1() / 2 - 1) + (size_t)lat.L1() * (lat.L2() / 2 - 1), nb = (size_t)lat.L1() * lat.L2();
                    const int nt_ = params.getInt("simulation", "nt");
                    std::cout << "      greenTau(r=0; tau = 0, beta/2, beta) " << ut[r0] << " " << ut[(size_t)(nt_ / 2) * nb + r0] << " " << ut[(size_t)nt_ * nb + r0] << '\n';
                }
                if (results) results->write_bin(rb);                   // measurements.accumulate -> saveToHDF5, include/measurementh5.h:253
            }
        }
        sim.download(greens[0]);
        const double local_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        const double acc = sim.acc_rate() / (n_bins * 2.0 * n_sweeps + 2.0 * n_therms);     // source/main.cpp:183
        const int total_sec = (int)local_time;
        std::cout << "DQMC measurement sweeps are finished in " << total_sec / 3600 << " hours " << (total_sec % 3600) / 60 << " minutes "
                  << total_sec % 60 << " seconds.\n"
                  << "Average acceptance rate = " << std::fixed << std::setprecision(4) << acc << '\n'
                  << "Max, Mean Precision Error = " << std::scientific << std::setprecision(4) << sim.max_err() << ", " << sim.mean_err() << '\n'
                  << "sweeps/s = " << std::fixed << std::setprecision(3) << (bin_sweeps > 0 ? bin_sweeps / local_time : 0.0) << '\n';
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "dqmc_driver: " << e.what() << std::endl;
        return 1;
    }
}
