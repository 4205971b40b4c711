"""Wall time of dqmc_inv_I_plus_ldr on a 256 x 256 LDR (split, R^-1 D solve, GJ solve: 8 panels + 8 updates, transposes; upload and
download included: a constant for A/B runs of two builds in one gpurun call).  usage: ab_solve.py lib.so [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(sys.argv[1], "dqmc_")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
M = np.random.default_rng(1).standard_normal((256, 256)) * np.exp(np.random.default_rng(2).uniform(-6, 6, 256))[None, :]
F = lib.to_ldr(M)
for _ in range(5): lib.inv_I_plus_ldr(F)
out = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(reps): lib.inv_I_plus_ldr(F)
    out.append((time.perf_counter() - t0) / reps * 1e6)
print(f"{os.path.basename(sys.argv[1]):40s} inv_I_plus_ldr us/call {np.round(out, 1)}")
