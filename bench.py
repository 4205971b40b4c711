#!/usr/bin/env python3
"""bench.py -- MC sweeps/s of the equal-time DQMC sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one Monte Carlo sweep (sweep_0_to_beta + sweep_beta_to_0 =
2*Ltau*N single-site proposals, source/main.cpp:156-157) of every Markov chain
this rank owns.  Workload: BASELINE.json configs[2], 16x16 Hubbard, U=8,
beta=8, Ltau=200, n_stab=10 -- one chain per GPU by default (the named
config); --chains-per-gpu C batches C independent chains into every kernel
launch.  Inputs are synthetic and resident in HBM before the timed region
(only the per-sweep random stream, 2.6 MB/chain, is uploaded asynchronously).

For N > 1 there is one rank per GPU (the reference's MPI ranks, source/main.cpp:20-37): either an
outer launcher (torch.distributed.run) has set RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, or -- plain
`python bench.py --gpus N` -- this process spawns the N ranks itself before it touches the GPU and relays
rank 0's JSON line.  Chains never communicate during sweeps (source/main.cpp:128-171), so there is no
data-path collective: scaling is weak, value = total sweeps / max time.  After the timed region the ranks
run two replica-exchange rounds over the library's RCCL transport (dqmc_replica_exchange_round; cfg 4's
wire) and report them under "replica_exchange" -- outside `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s achievable)
FP64_MFMA_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (AMD MI355X spec)


def gen_streams(model, rng, C):
    st = [model.random_stream(rng) for _ in range(C)]
    return tuple(np.stack([s[k] for s in st]) for k in range(3))


def parity_vs_cpu(model, orc, lib, device, config):
    """max|dG| of the HIP engine against the CPU oracle on identical fields and random streams (the metric's second half):
    init + one full sweep on the thermalised fixture of the config (the 1e-10 target is meaningful there).  The
    fixture also holds an independent numpy/scipy evaluation of the same sweep (tests/golden/make_golden.py), so the line
    carries the CPU-vs-CPU floor (oracle vs numpy) beside the GPU-vs-CPU figures."""
    from dqmc_amd import fixtures
    name = f"{config}_therm"
    if name in fixtures.NAMES:
        z, _, streams = fixtures.load(name)
        fields = z["fields"]; out = {"fields": f"thermalised fixture tests/golden/{name}.npz"}
    else:
        z = None; fields = model.random_fields(4711); out = {"fields": "iid random"}
        rng = np.random.default_rng(99); streams = (model.random_stream(rng), model.random_stream(rng))
    g = model.engine(lib, device=device); c = model.engine(orc)
    try:
        for e in (g, c):
            e.set_fields(fields); e.init()
        Gg, Gc = g.get_G(), c.get_G()
        out["max_abs_dG_init"] = float(np.abs(Gg - Gc).max()); out["max_abs_G"] = float(np.abs(Gc).max())
        if z is not None and "G0_rows" not in z.files:
            out["cpu_vs_cpu_floor_init"] = float(np.abs(Gc - z["G0"]).max()); out["max_abs_dG_init_vs_numpy"] = float(np.abs(Gg - z["G0"]).max())
        for e in (g, c):
            e.sweep_0_to_beta(*streams[0]); e.sweep_beta_to_0(*streams[1])
        sg, sc_ = g.stats(), c.stats()
        out["sweep_stats"] = {"gpu": {"acceptance": sg.n_accepted / max(1, sg.n_proposed), "max_wrap_err": sg.max_err, "mean_wrap_err": sg.mean_err},
                              "cpu": {"acceptance": sc_.n_accepted / max(1, sc_.n_proposed), "max_wrap_err": sc_.max_err, "mean_wrap_err": sc_.mean_err}}
        Gg, Gc = g.get_G(), c.get_G()
        out["max_abs_dG_after_sweep"] = float(np.abs(Gg - Gc).max())
        out["max_abs_G_after_sweep"] = float(np.abs(Gc).max())
        out["fields_identical_after_sweep"] = bool(np.array_equal(g.get_fields(), c.get_fields()))
        if z is not None and "G_after" in z.files:
            out["cpu_vs_cpu_floor_after_sweep"] = float(np.abs(Gc - z["G_after"]).max())
            out["max_abs_dG_after_sweep_vs_numpy"] = float(np.abs(Gg - z["G_after"]).max())
            out["fields_identical_to_numpy"] = bool(np.array_equal(g.get_fields(), z["fields_after"]))
        out["tolerance"] = "1e-10 absolute (BASELINE.json north_star), thermalised fields"
        out["within_tolerance"] = bool(out["max_abs_dG_init"] <= 1e-10 and out["max_abs_dG_after_sweep"] <= 1e-10)
        out["within_relative_tolerance"] = bool(out["max_abs_dG_init"] <= 1e-10 * max(1.0, out["max_abs_G"])
                                                and out["max_abs_dG_after_sweep"] <= 1e-10 * max(1.0, out["max_abs_G_after_sweep"]))
    finally:
        g.close(); c.close()
    return out


def cpu_baseline(model, seed, budget_s=20.0, lib=None, device=0, config="cfg3"):
    """The oracle (a port: the reference cannot be built here) timed on one host
    core on a bounded sample of the same workload."""
    from oracle import oracle
    orc = oracle()
    kind_blas = "lapack" if orc.set_backend("lapack") else "builtin"
    parity = None
    try:
        if lib is not None:
            parity = parity_vs_cpu(model, orc, lib, device, config)
        gold = os.path.join(ROOT, "tests", "golden", f"{config}_therm.npz")
        start_fields = (lambda k: np.load(gold)["fields"]) if os.path.exists(gold) else (lambda k: model.random_fields(seed + k))   # same start as the GPU run
        e = model.engine(orc); e.set_fields(start_fields(0)); e.init()
        rng = np.random.default_rng(seed)
        e.sweep_0_to_beta(*model.random_stream(rng)); e.sweep_beta_to_0(*model.random_stream(rng))   # warm-up sweep
        n = 0; t0 = time.perf_counter()
        while True:
            e.sweep_0_to_beta(*model.random_stream(rng)); e.sweep_beta_to_0(*model.random_stream(rng)); n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= 50:
                break
        backend = orc.backend()
        # the reference runs one Markov chain per MPI rank / host core (source/main.cpp:30-37): the same oracle on min(nproc, 8) cores at
        # once, one chain per thread (the library calls release the GIL, MKL is sequential) -- SURVEY.md 8(d)
        import threading
        K = max(1, min(8, os.cpu_count() or 1))
        counts = [0] * K; times = [0.0] * K
        def chain(i):
            ei = model.engine(orc); ei.set_fields(start_fields(1 + i)); ei.init()
            ri = np.random.default_rng(seed + 1 + i)
            ei.sweep_0_to_beta(*model.random_stream(ri)); ei.sweep_beta_to_0(*model.random_stream(ri))
            t_0 = time.perf_counter()
            while True:
                ei.sweep_0_to_beta(*model.random_stream(ri)); ei.sweep_beta_to_0(*model.random_stream(ri)); counts[i] += 1
                times[i] = time.perf_counter() - t_0
                if times[i] > 0.5 * budget_s or counts[i] >= 50:
                    break
            ei.close()
        multi = None
        if K > 1 and budget_s > 0:
            th = [threading.Thread(target=chain, args=(i,)) for i in range(K)]
            for t_ in th: t_.start()
            for t_ in th: t_.join()
            multi = {"cores": K, "aggregate": sum(c / t_ for c, t_ in zip(counts, times) if t_ > 0), "unit": "sweeps/s",
                     "per_chain": float(np.mean([c / t_ for c, t_ in zip(counts, times) if t_ > 0])),
                     "sample": f"{K} independent chains, one per thread, {sum(counts)} sweeps in total"}
    finally:
        orc.set_backend("builtin")
    out = {"value": n / dt, "unit": "sweeps/s", "cores": 1, "kind": "port",
           "sample": f"{n} sweeps of 1 chain after 1 warm-up sweep, single thread, {backend}",
           "nproc": os.cpu_count(), "flags": "g++ -O3 -march=native -fno-math-errno (no -ffast-math), oracle/Makefile"}
    if multi is not None:
        out["all_cores"] = multi
    if parity is not None:
        out["max_dG_vs_cpu"] = parity
    return out


def pmc_traffic(config, chains):
    """Measured HBM bytes per launch of the dominant kernel (counter passes committed under profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "r04_pmc_slice_kernel.json")
    if config != "cfg3" or chains != 1 or not os.path.exists(path):
        return None
    return float(json.load(open(path))["traffic_bytes_per_launch"])


def spawn_ranks(n: int, argv, extra_env=None, timeout=None) -> int:
    """Start n fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and wait for them;
    their stdout / stderr pass through.  Returns the largest exit code.  Called before the parent makes any GPU call."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        for p_ in procs:
            rc = max(rc, abs(p_.wait(timeout=timeout)))
    except BaseException:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()                           # exactly the processes started here
        raise
    return rc


def replica_exchange_check(lib, eng, d, rounds=2, timeout_s=120.0):
    """Two rounds of dqmc_replica_exchange_round among the ranks over the RCCL transport (one rank per GPU), after the
    timed region: the id travels over torch.distributed, the swaps HBM to HBM.  Guarded by a watchdog: a communicator that
    never forms is reported, not waited for."""
    import threading
    import torch.distributed as dist
    out = {"transport": "rccl", "rounds": rounds}
    done = threading.Event()

    def body():
        try:
            ids = [lib.comm_unique_id() if d.rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = lib.comm_rccl(ids[0], d.world, d.rank, d.local_rank)
            comm.barrier()
            rng = np.random.default_rng(4000 + d.rank)
            res = []
            t0 = time.perf_counter()
            for attempt in range(1, rounds + 1):
                r = comm.exchange_round(eng, attempt, float(rng.random()))
                res.append({"partner": r.partner, "decider": r.decider, "accepted": r.accepted, "deltaS": r.deltaS})
            comm.barrier()
            out["ms_per_round"] = 1e3 * (time.perf_counter() - t0) / rounds
            out["rank0"] = res
            acc = comm.allreduce_sum([float(sum(x["accepted"] for x in res))])
            out["accepted_pair_ends"] = int(acc[0]); out["pair_ends"] = d.world * rounds
            comm.close()
        except Exception as e:                      # noqa: BLE001
            out["error"] = repr(e)
        done.set()
    th = threading.Thread(target=body, daemon=True); th.start()
    if not done.wait(timeout_s):
        out["error"] = f"no completion within {timeout_s:.0f} s"
        out["hung"] = True
    return out


def cfg5_block(lib, device):
    """BASELINE.json configs[4] (24x24 Hubbard, beta = 10, Ltau = 400, n_stab = 10) in the driver-observed line: from the seeded i.i.d.
    fields of the golden fixture, init + 1 warm-up sweep + 2 timed sweeps of one chain.  Reported beside the headline, never as
    `value`.  The roofline row is the same as the headline's: 16 N^2 algorithmic bytes per accepted flip over the HIP-event time of
    the local-update launches (slice_sm_kernel: the sub-matrix walk + flush roles)."""
    import dqmc_amd
    m = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS["cfg5"])
    gold = os.path.join(ROOT, "tests", "golden", "cfg5_random_init.npz")
    fields = np.load(gold)["fields"] if os.path.exists(gold) else m.random_fields(55)
    e = m.engine(lib, device=device); e.set_fields(fields); e.init()
    rng = np.random.default_rng(555)

    def sw():
        e.sweep_0_to_beta(*m.random_stream(rng)); e.sweep_beta_to_0(*m.random_stream(rng))
    sw(); e.sync()
    s0 = e.stats()
    e.set_profiling(True); e.update_kernel_time()
    steps = 2
    t0 = time.perf_counter()
    for _ in range(steps):
        sw()
    e.sync()
    dt = time.perf_counter() - t0
    ms, launches, acc = e.update_kernel_time(); e.set_profiling(False)
    s1 = e.stats()
    n = m.n
    bpl = 16.0 * n * n * acc / max(1, launches); avg = (ms * 1e-3) / max(1, launches)
    rec = {"workload": f"cfg5: {m.L1}x{m.L2} Hubbard U={m.U} beta={m.beta} Ltau={m.nt} n_stab={m.n_stab}, one chain, start = i.i.d. fixture fields + 1 warm-up sweep",
           "value": steps / dt, "unit": "sweeps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "acceptance": (s1.n_accepted - s0.n_accepted) / max(1, s1.n_proposed - s0.n_proposed), "max_wrap_err": s1.max_err,
           "slice_path": e.slice_path(),
           "roofline": {"kernel": "slice_sm_kernel (sub-matrix walk + flush roles)", "bound": "hbm", "algorithmic_bytes_per_launch": bpl, "avg_launch_ms": avg * 1e3,
                        "achieved": bpl / avg / 1e9 if avg > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (bpl / avg / 1e9 / HBM_PEAK_GBS) if avg > 0 else 0.0,
                        "launches": launches, "accepted": acc}}
    e.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--chains-per-gpu", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batched-chains", type=int, default=128,
                    help="also time this many chains batched on one GPU (N=1 only; 0 disables); reported beside `value`, never as `value`")
    ap.add_argument("--batched-engines", type=int, default=1,
                    help="engines (HIP streams, one host thread each) the batched chains are spread over; >1 only pays with one PROCESS per engine "
                         "(measured: 4 processes x 32 chains 303 sweeps/s, 1 x 128 chains 272, 4 threads x 32 chains 226)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="functional rehearsal of the N > 1 path on a box with fewer GPUs than ranks: gloo rendezvous, ranks share devices "
                         "(local_rank %% device_count); the timing is meaningless")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous, barrier and max-over-ranks only (gloo, no engine, no GPU): the CPU test of the N > 1 launch path")
    ap.add_argument("--checkerboard", action="store_true",
                    help="opt-in: checkerboard break-up of exp(-dtau K) (dqmc_set_checkerboard) instead of the reference's dense exponential -- a "
                         "different discretisation, reported with \"kinetic\": \"checkerboard\" and never comparable with the headline line")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the BASELINE.json configs[4] block (24x24, Ltau = 400: init + 1 warm-up + 2 timed sweeps, ~5 s) of the N = 1 line")
    ap.add_argument("--no-replica-exchange", action="store_true", help="skip the RCCL replica-exchange rounds after the timed region (N > 1)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It makes no GPU call: the device count comes from
        # torch.cuda.device_count(), which does not initialise the runtime on this image
        if not (args.rehearse_shared_gpu or args.launch_check):
            import torch
            have = torch.cuda.device_count()
            if have < args.gpus:
                print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible (use --rehearse-shared-gpu for a functional rehearsal)", file=sys.stderr)
                sys.exit(2)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={env_world} set by the launcher", file=sys.stderr)
        sys.exit(2)

    import torch
    from dqmc_amd.launch import barrier as dist_barrier, dist_init, finalize, max_over_ranks
    d = dist_init("gloo" if (args.rehearse_shared_gpu or args.launch_check) else None)      # WORLD_SIZE > 1: backend "nccl" (= RCCL) on GPUs
    rank, world, local_rank = d.rank, d.world, d.local_rank
    if args.launch_check:
        dist_barrier(d)
        t = max_over_ranks(d, 1.0 + rank)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "max_over_ranks": t, "backend": d.backend}), flush=True)
        finalize(d)
        return
    if not args.rehearse_shared_gpu and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        sys.exit(2)

    import dqmc_amd
    lib = dqmc_amd.lib()
    if lib.device_count() == 0:
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU fallback")
    model = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS[args.config])
    C = args.chains_per_gpu
    eng = model.engine(lib, device=(local_rank % lib.device_count()) if args.rehearse_shared_gpu else local_rank, n_chains=C)
    # thermalise before timing.  When the config has a thermalised fixture (fields after CPU sweeps from the
    # seeded i.i.d. start, tests/golden/make_golden.py) every chain starts from it and diverges at once through its own random
    # stream; otherwise i.i.d. fields with seed 12345 + chain id, and --warmup should then be >= 20
    gold = os.path.join(ROOT, "tests", "golden", f"{args.config}_therm.npz")
    if os.path.exists(gold):
        f0 = np.load(gold)["fields"]
        fields = np.stack([f0 for _ in range(C)]); start = "thermalised fixture tests/golden/%s_therm.npz + %d warm-up sweeps" % (args.config, args.warmup)
    else:
        fields = np.stack([model.random_fields(12345 + rank * C + c) for c in range(C)]); start = "iid random fields (seed 12345 + chain) + %d warm-up sweeps" % args.warmup
    if args.checkerboard:
        eng.set_checkerboard(*model.checkerboard()); args.no_cpu_baseline = True; args.batched_chains = 0
    eng.set_fields(fields); eng.init()
    rng = np.random.default_rng(777 + rank)

    def barrier():
        dist_barrier(d)

    def sweep():
        eng.sweep_0_to_beta(*gen_streams(model, rng, C))
        eng.sweep_beta_to_0(*gen_streams(model, rng, C))

    for _ in range(args.warmup):
        sweep()
    eng.sync()
    st0 = eng.stats()
    eng.set_profiling(True); eng.update_kernel_time()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sweep()
    eng.sync()
    barrier()
    dt = time.perf_counter() - t0
    upd_ms, upd_launches, upd_acc = eng.update_kernel_time()
    eng.set_profiling(False)
    st1 = eng.stats()
    dt = max_over_ranks(d, dt)

    if rank == 0:
        n = model.n
        total_sweeps = world * C * args.steps
        s0 = st0[0] if C > 1 or isinstance(st0, list) else st0
        s1 = st1[0] if C > 1 or isinstance(st1, list) else st1
        acc_rate = (s1.n_accepted - s0.n_accepted) / max(1, s1.n_proposed - s0.n_proposed)
        # rank-1 update roofline: 16*N^2 algorithmic bytes per ACCEPTED proposal (read + write G once,
        # SURVEY.md 8d); one launch = the local update of one time slice (all chains of this rank)
        bytes_per_launch = 16.0 * n * n * upd_acc / max(1, upd_launches)
        avg_launch_s = (upd_ms * 1e-3) / max(1, upd_launches)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # GEMM chain: G_alg = 6*Ltau + 8*n_stack - 10 dense GEMMs of 2N^3 flop per sweep
        g_alg = 6 * model.nt + 8 * model.n_stack - 10
        out = {
            "metric": "MC sweeps/sec, 2D Hubbard N=256 Ltau=200", "value": total_sweeps / dt, "unit": "sweeps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {model.L1}x{model.L2} Hubbard U={model.U} beta={model.beta} Ltau={model.nt} n_stab={model.n_stab} t=1 mu=-0.1",
                       "chains_per_gpu": C, "n_chains_total": world * C, "parallelism": f"independent chains, {world} gpu x {C} chain"},
            "start": start, "kinetic": "checkerboard" if args.checkerboard else "dense exp(-dtau K) (the reference's)",
            "acceptance": acc_rate, "max_wrap_err": s1.max_err, "mean_wrap_err": s1.mean_err,
            "roofline": {"kernel": "local update of one time slice (" + ("slice_kernel: delayed-update walk + flush roles" if n <= 256 else
                                                                         "slice_sm_kernel: sub-matrix walk + flush roles") + ")", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # HBM-side bytes per launch from rocprofv3 PMC passes of this round's binary on the same start state (FETCH_SIZE x2
                         # gfx950 correction + WRITE_SIZE, separate passes; profiles/r04_pmc_slice_kernel.json names the CSVs and the command);
                         # cfg3, single chain, single-launch path only
                         "traffic": pmc_traffic(args.config, C),
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_launch_s * 1e3,
                         "launches": upd_launches, "accepted": upd_acc,
                         "time_share_of_sweep": (upd_ms * 1e-3) / dt},
            "gemm_chain": {"algorithmic_gemms_per_sweep": g_alg, "flop_per_sweep": g_alg * 2.0 * n ** 3,
                           "peak_tflops": FP64_MFMA_PEAK_TFLOPS},
        }
        # GEMM chain in situ: 200 wraps (B G B^-1 then B^-1 G B: 4 GEMMs of 2N^3 flop, G returns to itself up to rounding) enqueued
        # back to back on the engine's stream, the way a sweep issues them; wall time / 800 = time per GEMM including its launch gap
        n_pairs = 200
        eng.sync(); tg0 = time.perf_counter()
        for _ in range(n_pairs):
            eng.wrap_forward(0); eng.wrap_backward(0)
        eng.sync(); tg = (time.perf_counter() - tg0) / (4 * n_pairs)
        gflop = 2.0 * n ** 3 * C
        out["gemm_chain"].update({"avg_gemm_us": tg * 1e6, "achieved_tflops": gflop / tg / 1e12, "frac": gflop / tg / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                  "bound": "mfma", "note": "checkerboard: these are the pair kernels, not GEMMs" if args.checkerboard else "one 256^3 GEMM is launch/latency bound; the batched engines run the same chain at 41 TFLOP/s (DESIGN.md 4)"})
        if world == 1 and args.batched_chains > 1 and C == 1:
            # throughput mode: the same kernels advance B independent chains per launch (blockIdx.y = chain), optionally spread
            # over E engines (own HIP stream and host thread each).  The random streams are drawn before the clock starts: with
            # 128 chains numpy needs as long to draw them as the GPU needs to consume them
            import threading
            B, E = args.batched_chains, max(1, args.batched_engines)
            per = [B // E + (1 if e < B % E else 0) for e in range(E)]
            engs = []
            for e, nb_e in enumerate(per):
                ee = model.engine(lib, device=local_rank, n_chains=nb_e)
                ee.set_fields(np.stack([model.random_fields(50000 + 1000 * e + c) for c in range(nb_e)])); ee.init()
                engs.append((ee, np.random.default_rng(4242 + e), nb_e))
            def run(ee, streams):
                for (fw, bw) in streams:                    # the library calls release the GIL; the streams are drawn beforehand
                    ee.sweep_0_to_beta(*fw); ee.sweep_beta_to_0(*bw)
                ee.sync()
            def run_all(reps):
                work = [(ee, [(gen_streams(model, rr, nb_e), gen_streams(model, rr, nb_e)) for _ in range(reps)]) for (ee, rr, nb_e) in engs]
                th = [threading.Thread(target=run, args=w_) for w_ in work]
                t_start = time.perf_counter()
                for t_ in th: t_.start()
                for t_ in th: t_.join()
                return time.perf_counter() - t_start
            run_all(1)
            nb = 2
            tb = run_all(nb)
            out["batched"] = {"chains_per_gpu": B, "engines": E, "value": B * nb / tb, "unit": "sweeps/s", "ms_per_step": 1e3 * tb / nb, "steps": nb,
                              "note": "aggregate over independent chains sharing kernel launches; not the headline config"}
            for (ee, _, _) in engs: ee.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, 12345, args.cpu_budget, lib, local_rank, args.config)
        if args.rehearse_shared_gpu:
            out["rehearsal"] = True                  # ranks share devices: the launch path is exercised, the number means nothing
    rex = None
    if world > 1 and C == 1 and not args.no_replica_exchange and not args.rehearse_shared_gpu:
        rex = replica_exchange_check(lib, eng, d)
    if rank == 0 and world == 1 and C == 1 and args.config == "cfg3" and not args.checkerboard and not args.no_cfg5:
        eng.close()                                   # hands back its CU reservation: the cfg-5 engine takes the persistent slice kernel as it does on its own
        out["cfg5"] = cfg5_block(lib, local_rank)
    if rank == 0:
        if rex is not None:
            out["replica_exchange"] = rex
        print(json.dumps(out), flush=True)
    if rex is not None and rex.get("hung"):
        # a communicator that never formed cannot be torn down either: leave without teardown, but NOT with success -- the JSON line
        # (already printed, "hung": true) carries the measurement, exit code 3 tells the driver that the post-timing RCCL check hung
        sys.stdout.flush(); sys.stderr.flush(); os._exit(3)
    eng.close()
    finalize(d)


if __name__ == "__main__":
    main()
