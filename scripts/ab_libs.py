"""Timing of one build of libdqmc_hip.so on a fixed thermalised cfg-3 state and fixed random streams (A/B runs: call it once
per build, in separate processes -- the library is loaded RTLD_GLOBAL, two builds in one process would interpose each other):
python scripts/ab_libs.py lib.so [sweeps]   -> ms per sweep and us per local-update launch, three repetitions"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dqmc_amd
from dqmc_amd.abi import DqmcLib
paths = sys.argv[1:2]
nsw = int(sys.argv[2]) if len(sys.argv) > 2 else 2
the_lib = DqmcLib(paths[0], "dqmc_")
m = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS[os.environ.get("AB_CFG", "cfg3")])      # AB_CFG=cfg5: the N = 576 path
e0 = m.engine(the_lib); e0.set_fields(m.random_fields(3)); e0.init()
rng = np.random.default_rng(0)
for _ in range(3):
    e0.sweep_0_to_beta(*m.random_stream(rng)); e0.sweep_beta_to_0(*m.random_stream(rng))
fields = e0.get_fields(); del e0
streams = [m.random_stream(np.random.default_rng(100 + i)) for i in range(2 * nsw)]
if os.environ.get("AB_NOACC"):      # u = 1: no proposal is ever accepted -> the fixed cost of a slice
    streams = [(p_, k_, np.ones_like(u_)) for (p_, k_, u_) in streams]
res = {p: [] for p in paths}
for rep in range(3):
    for p in paths:
        lib = the_lib
        e = m.engine(lib); e.set_fields(fields); e.init()
        e.sweep_0_to_beta(*streams[0]); e.sweep_beta_to_0(*streams[1]); e.sync()          # warm-up
        e.set_profiling(True); e.update_kernel_time()
        t0 = time.perf_counter()
        for i in range(nsw):
            e.sweep_0_to_beta(*streams[2 * i]); e.sweep_beta_to_0(*streams[2 * i + 1])
        e.sync()
        dt = time.perf_counter() - t0
        ms, launches, acc = e.update_kernel_time()
        res[p].append((1e3 * dt / nsw, 1e3 * ms / launches, acc / launches))
        del e
for p in paths:
    r = np.array(res[p])
    print("%-48s ms/sweep %s  us/slice %s  acc/slice %.1f" % (os.path.basename(p), np.round(r[:, 0], 1), np.round(r[:, 1], 1), r[0, 2]), flush=True)
