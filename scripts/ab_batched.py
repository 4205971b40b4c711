"""Timing of one build of libdqmc_hip.so on a BATCHED cfg-3 engine (C chains in every launch), for A/B runs in one gpurun call:
python scripts/ab_batched.py lib.so [chains]   -> ms per step (one sweep of all chains) and aggregate sweeps/s, three repetitions"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dqmc_amd
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(sys.argv[1], "dqmc_")
C = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS["cfg3"])
f0 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfg3_therm.npz"))["fields"]
e = m.engine(lib, n_chains=C); e.set_fields(np.stack([f0] * C)); e.init()
rng = np.random.default_rng(5)
def streams():
    s = [m.random_stream(rng) for _ in range(C)]
    return tuple(np.stack([x[i] for x in s]) for i in range(3))
def sweep():
    e.sweep_0_to_beta(*streams()); e.sweep_beta_to_0(*streams())
sweep(); e.sync()
res = []
for rep in range(3):
    st = [(streams(), streams())]
    t0 = time.perf_counter()
    for a, b in st:
        e.sweep_0_to_beta(*a); e.sweep_beta_to_0(*b)
    e.sync(); res.append(time.perf_counter() - t0)
print(f"{os.path.basename(sys.argv[1]):40s} chains {C}  ms/step {np.round(np.array(res) * 1e3, 1)}  aggregate sweeps/s {C / min(res):.1f}")
