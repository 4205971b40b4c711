// main.cpp -- a driver with the loop shape of the reference's source/main.cpp:14-214 on top of the host facade: read
// ./parameters.in (or argv[1]), build Lattice / AttractiveHubbard / DQMC, thermalise, sweep (with replica exchange every
// `sweep_steps` sweeps when [ParallelTempering] enabled = true), report time, acceptance, wrap error and exchange rate.
//
//   dqmc_driver [parameters.in [device [seed [rank]]]]
//
// One rank = one Markov chain on one GPU (the reference's MPI rank, source/main.cpp:20-37).  Ranks are
//   * processes started by any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torchrun), PMI_RANK / PMI_SIZE (mpiexec),
//     OMPI_COMM_WORLD_RANK / _SIZE / _LOCAL_RANK or DQMC_RANK / DQMC_WORLD_SIZE / DQMC_LOCAL_RANK: the ranks meet in an RCCL
//     communicator (dqmc_comm_create_rccl; the 128-byte id travels through the file $DQMC_RENDEZVOUS_FILE, default
//     results/.dqmc_rccl_id.<parent pid>), replica swaps go GPU to GPU over xGMI; or
//   * threads of this one process when parallel tempering is enabled and no launcher is present: one replica per beta,
//     replica r on device r % device_count, swaps through the in-process hub (update::InProcessHub).
// Observables are accumulated per bin on the device, printed, and written to results/data_<rank>.h5 in the
// reference's layout (results_h5.hpp; DQMC_NO_HDF5=1 or a missing libhdf5 turns the file off with a warning).
#include "dqmc_host.hpp"
#include "results_h5.hpp"

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <memory>
#include <thread>

namespace {

struct World { int rank = 0, world = 1, local_rank = 0; bool from_env = false; };

World world_from_env() {
    static const char* const sets[][3] = {{"DQMC_RANK", "DQMC_WORLD_SIZE", "DQMC_LOCAL_RANK"}, {"RANK", "WORLD_SIZE", "LOCAL_RANK"},
                                          {"OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK"},
                                          {"PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID"}, {"SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID"}};
    World w;
    for (const auto& s : sets) {
        const char* r = std::getenv(s[0]); const char* n = std::getenv(s[1]);
        if (!r || !n) continue;
        w.rank = std::atoi(r); w.world = std::atoi(n); w.from_env = true;
        const char* l = std::getenv(s[2]);
        w.local_rank = l ? std::atoi(l) : w.rank;
        break;
    }
    return w;
}

// rank 0 publishes the RCCL id in a file, the others wait for it (no MPI in the picture)
dqmc_comm* rccl_world(const World& w, int device) {
    std::string path;
    if (const char* p = std::getenv("DQMC_RENDEZVOUS_FILE")) path = p;
    else {
        // one file per RUN: the launcher's pid plus, when the launcher exports one, its per-run nonce (torchrun: TORCHELASTIC_RUN_ID; mpirun:
        // OMPI_MCA_ess_base_jobid / PMI_JOBID; or DQMC_RUN_ID set by hand), so that a file a crashed earlier run of a recycled pid left
        // behind is never this run's
        std::string nonce;
        for (const char* k : {"DQMC_RUN_ID", "TORCHELASTIC_RUN_ID", "OMPI_MCA_ess_base_jobid", "PMI_JOBID", "SLURM_JOB_ID"})
            if (const char* v = std::getenv(k)) { nonce = std::string(".") + v; break; }
        mkdir("results", 0755); path = "results/.dqmc_rccl_id." + std::to_string((long)getppid()) + nonce;
    }
    char id[DQMC_UNIQUE_ID_BYTES];
    if (w.rank == 0) {
        dqmc_detail::check(dqmc_comm_unique_id(id));
        std::remove(path.c_str());                                  // a stale leftover of an earlier run
        const std::string tmp = path + ".tmp";
        { std::ofstream f(tmp, std::ios::binary); f.write(id, sizeof(id)); }
        if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot publish the RCCL id at " + path);
    } else {
        // a file left behind by a crashed run of the same launcher pid must not be taken for this run's: only a file written after
        // this process started (minus a margin for launcher skew) counts
        const time_t t_start = time(nullptr) - 120;
        bool ok = false;
        for (int tries = 0; tries < 6000 && !ok; ++tries) {
            struct stat stt;
            std::ifstream f(path, std::ios::binary);
            if (stat(path.c_str(), &stt) == 0 && stt.st_mtime >= t_start && f.is_open() && f.read(id, sizeof(id)) && f.gcount() == (std::streamsize)sizeof(id)) ok = true;
            else std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!ok) throw std::runtime_error("timed out waiting for the RCCL id at " + path);
    }
    dqmc_comm* c = nullptr;
    dqmc_detail::check(dqmc_comm_create_rccl(&c, id, w.world, w.rank, device));
    dqmc_detail::check(dqmc_comm_barrier(c));
    if (w.rank == 0) std::remove(path.c_str());
    return c;
}

std::mutex g_print_mu, g_h5_mu;      // in-process replicas share stdout and libhdf5 (not thread-safe)

struct RunConfig { std::string pfile; unsigned seed; };

// the body of source/main.cpp:36-211 for one rank
int run_rank(const RunConfig& rc, int rank, int world_size, int device, dqmc_comm* comm) {
    const int master = 0;
    utility::parameters params(rc.pfile);
    utility::random rng(rc.seed + (unsigned)rank);                                             // source/main.cpp:37 (time(nullptr) + rank)
    const bool pt_enabled = params.getBool("ParallelTempering", "enabled", false);
    double my_beta; int exchange_step = 0, exchange_attempt = 0, exchange_accepted = 0;
    if (pt_enabled) {                                                                          // source/main.cpp:47-67 (the checks ran in main())
        const std::vector<double> betas = params.getDoubleVector("ParallelTempering", "betas");
        my_beta = betas.at(rank);
        exchange_step = params.getInt("ParallelTempering", "sweep_steps");
        if (exchange_step <= 0) throw std::runtime_error("[ParallelTempering] sweep_steps must be positive");
    } else my_beta = params.getDouble("simulation", "beta");
    const int n_sweeps = params.getInt("simulation", "n_sweeps"), n_therms = params.getInt("simulation", "n_therms"), n_bins = params.getInt("simulation", "n_bins");

    Lattice lat(params, {1.0, 0.0}, {0.0, 1.0}, {{0.0, 0.0}});
    if (rank == master || world_size == 1) lat.save_info("results/info");      // world_size 1: `rank` is only the label of an independent chain
    AttractiveHubbard model(params, lat, rng, my_beta);
    const int n_flavor = model.n_flavor();
    DQMC sim(params, model, device);

    std::vector<LDRStack> propagation_stacks(n_flavor);
    std::vector<GF> greens(n_flavor);
    const bool symmetric = params.getBool("simulation", "symmetric", false);                   // source/main.cpp:75
    std::vector<GF> greens_symm(symmetric ? n_flavor : 0);                                     // source/main.cpp:108-113
    for (int flv = 0; flv < n_flavor; flv++) {
        propagation_stacks[flv] = sim.init_stacks(flv);
        greens[flv] = sim.init_greenfunctions(propagation_stacks[flv]);
    }
    {
        std::lock_guard<std::mutex> lk(g_print_mu);
        if (rank == master) std::cout << (pt_enabled ? "Parallel Tempering enabled" : "Standard DQMC run (Parallel Tempering disabled)") << ", " << world_size
                                      << " rank(s), transport " << (comm ? dqmc_comm_transport(comm) : "none") << ", backend " << dqmc_backend() << "\n";
        std::cout << "rank " << rank << ": device " << device << ", beta " << my_beta << ", log det M = " << std::setprecision(12) << greens[0].log_det_M << "\n";
    }

    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n_therms; ++i) { sim.sweep_0_to_beta(greens, propagation_stacks); sim.sweep_beta_to_0(greens, propagation_stacks); }
    sim.download(greens[0]);
    const double dt_therm = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rank == master) { std::lock_guard<std::mutex> lk(g_print_mu); std::cout << "Thermalization done in " << dt_therm << " seconds\n"; }

    std::unique_ptr<ResultsWriter> results;
    if (!std::getenv("DQMC_NO_HDF5")) {
        std::lock_guard<std::mutex> lk(g_h5_mu);
        try { results.reset(new ResultsWriter("results", rank, lat.L1(), lat.L2(), lat.a1(), lat.a2(), lat.k_points())); }
        catch (const std::exception& e) { std::cerr << "warning: no HDF5 output (" << e.what() << ")\n"; }
    }
    const long bin_sweeps = (long)n_bins * n_sweeps;
    double warned_err = 0.0;
    const auto t1 = std::chrono::steady_clock::now();
    for (long isweep = 1; isweep <= bin_sweeps; ++isweep) {
        if (pt_enabled && (isweep % exchange_step == 0)) {                                     // source/main.cpp:146-153
            dqmc_detail::check(dqmc_comm_barrier(comm));
            update::replica_exchange(rank, world_size, rng, exchange_attempt, exchange_accepted, model, sim, greens, propagation_stacks, comm);
        }
        sim.sweep_0_to_beta(greens, propagation_stacks);
        sim.sweep_beta_to_0(greens, propagation_stacks);
        sim.sweep_unequalTime(greens, propagation_stacks);
        if (symmetric) sim.half_warp(greens_symm, greens);          // source/main.cpp:161-163 (the reference measures `greens` all the same, :165)
        sim.measure(lat);                                          // measurements.measure(greens, lat), source/main.cpp:165 -- on the device, asynchronous
        sim.measure_unequal(lat);                                  // the dynamical observables of the same call, when isMeasureUnequalTime
        if (isweep % n_sweeps == 0) {                              // measurements.accumulate(lat), :167-169: one bin done
            const DQMC::EqualTime bin = sim.fetch_bin(lat);
            // the reference warns at every stabilisation whose wrap-vs-stabilised difference exceeds 1e-6 (source/dqmc.cpp:390-393,
            // :450-453); the engine folds the errors on the device, so the check runs where the host synchronises anyway: once per bin
            if (const double me = sim.max_err(); me > 1e-6 && me > warned_err) {
                std::lock_guard<std::mutex> lk(g_print_mu);
                std::cerr << "WARNING: GF precision > 1e-6 (rank " << rank << "). Reduce n_stab or increasing nt. Error: " << me << std::endl;
                warned_err = me;
            }
            ResultsBin rb; rb.density = bin.density; rb.doubleOcc = bin.doubleOcc; rb.swave = bin.swave; rb.densityCorr_r = bin.densityCorr_r;
            std::vector<double> ut;
            const int nt_ = params.getInt("simulation", "nt");
            if (sim.isUnequalTime()) { long long nu = 0; ut = sim.fetch_unequal_bin(lat, &nu); rb.unequal_r = ut; rb.n_tau = nt_ + 1; }
            {
                std::lock_guard<std::mutex> lk(g_print_mu);
                const size_t r0 = (size_t)(lat.L1() / 2 - 1) + (size_t)lat.L1() * (lat.L2() / 2 - 1), nb = (size_t)lat.L1() * lat.L2();
                std::cout << "rank " << rank << " bin " << isweep / n_sweeps << " (" << bin.n << " sweeps): density " << std::setprecision(8) << bin.density
                          << "  doubleOcc " << bin.doubleOcc << "  swave " << bin.swave << "  densityCorr(r=0) " << bin.densityCorr_r[r0] << '\n';
                if (!ut.empty()) std::cout << "      greenTau(r=0; tau = 0, beta/2, beta) " << ut[r0] << " " << ut[(size_t)(nt_ / 2) * nb + r0] << " " << ut[(size_t)nt_ * nb + r0] << '\n';
            }
            if (results) { std::lock_guard<std::mutex> lk(g_h5_mu); results->write_bin(rb); }   // measurements.accumulate -> saveToHDF5, include/measurementh5.h:253
        }
    }
    sim.download(greens[0]);
    const double local_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    const double local_acc_rate = sim.acc_rate() / (n_bins * 2.0 * n_sweeps + 2.0 * n_therms);   // source/main.cpp:183
    double sums[2] = {local_time, local_acc_rate};                                            // the two MPI_Reduce(SUM), :186-187
    if (comm) dqmc_detail::check(dqmc_comm_allreduce_sum(comm, sums, 2));
    {
        std::lock_guard<std::mutex> lk(g_print_mu);
        std::cout << "rank " << rank << ": Max, Mean Precision Error = " << std::scientific << std::setprecision(4) << sim.max_err() << ", " << sim.mean_err()
                  << std::fixed << ", acceptance " << local_acc_rate << '\n';
    }
    if (rank == master) {
        std::lock_guard<std::mutex> lk(g_print_mu);
        const int total_sec = (int)local_time;
        std::cout << "DQMC measurement sweeps are finished in " << total_sec / 3600 << " hours " << (total_sec % 3600) / 60 << " minutes "
                  << total_sec % 60 << " seconds.\n"
                  << "Average acceptance rate = " << std::fixed << std::setprecision(4) << sums[1] / world_size << '\n'
                  << "sweeps/s = " << std::fixed << std::setprecision(3) << (bin_sweeps > 0 ? world_size * bin_sweeps / (sums[0] / world_size) : 0.0)
                  << " (" << world_size << " chain(s))\n";
        if (pt_enabled)                                                                        // source/main.cpp:203-208
            std::cout << "Parallel tempering exchange rate = " << std::fixed << std::setprecision(4)
                      << (exchange_attempt ? static_cast<double>(exchange_accepted) / exchange_attempt : 0.0) << " (" << exchange_accepted << "/" << exchange_attempt << ")\n";
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        RunConfig rc;
        rc.pfile = argc > 1 ? argv[1] : "parameters.in";
        const int device_arg = argc > 2 ? std::atoi(argv[2]) : -1;
        rc.seed = argc > 3 ? (unsigned)std::strtoul(argv[3], nullptr, 10) : (unsigned)std::time(nullptr);   // source/main.cpp:37
        World w = world_from_env();
        if (argc > 4) { w.rank = std::atoi(argv[4]); }                                                       // rank label of an independent chain (results/data_<rank>.h5)
        utility::parameters params(rc.pfile);
        const bool pt_enabled = params.getBool("ParallelTempering", "enabled", false);
        const int n_dev = dqmc_device_count();
        if (n_dev == 0) throw std::runtime_error("no HIP device available: this program requires a gfx950 GPU");
        int n_replicas = 1;
        if (pt_enabled) {
            n_replicas = (int)params.getDoubleVector("ParallelTempering", "betas").size();
            const int ranks = w.from_env ? w.world : n_replicas;                                            // no launcher: one thread per beta
            if (n_replicas != ranks) {                                                                       // source/main.cpp:52-57
                std::cerr << "ERROR: The number of betas (" << n_replicas << ") in parameters.in must match the number of MPI processes (" << ranks << ")." << std::endl;
                return 1;
            }
            if (ranks % 2 != 0) {                                                                            // :58-62
                std::cerr << "ERROR: currently number of processor ( nprocs = " << ranks << ") need to be even for replica exchange" << std::endl;
                return 1;
            }
        }
        if (pt_enabled && !w.from_env) {
            // single process: the replicas are threads, one engine each, spread over the visible GPUs
            update::InProcessHub hub(n_replicas);
            std::vector<std::thread> th; std::vector<int> rcs(n_replicas, 0); std::vector<std::string> errs(n_replicas);
            for (int r = 0; r < n_replicas; ++r) th.emplace_back([&, r] {
                dqmc_comm* comm = nullptr;
                try { comm = hub.make_comm(r); rcs[r] = run_rank(rc, r, n_replicas, device_arg >= 0 ? device_arg : r % n_dev, comm); }
                catch (const std::exception& e) { errs[r] = e.what(); rcs[r] = 1; hub.abort(); }
                dqmc_comm_destroy(comm);
            });
            for (auto& t : th) t.join();
            for (int r = 0; r < n_replicas; ++r) if (rcs[r]) { std::cerr << "dqmc_driver: replica " << r << ": " << errs[r] << std::endl; return 1; }
            return 0;
        }
        const int device = device_arg >= 0 ? device_arg : w.local_rank % n_dev;
        dqmc_comm* comm = (w.from_env && w.world > 1) ? rccl_world(w, device) : nullptr;
        const int rcode = run_rank(rc, w.rank, w.from_env ? w.world : 1, device, comm);
        dqmc_comm_destroy(comm);
        return rcode;
    } catch (const std::exception& e) {
        std::cerr << "dqmc_driver: " << e.what() << std::endl;
        return 1;
    }
}
