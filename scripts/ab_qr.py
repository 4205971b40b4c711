"""Wall time of dqmc_to_ldr on a graded 256 x 256 matrix (QRCP + form-Q + assembly, upload and download included: a constant for
A/B runs of two builds in one gpurun call).  usage: ab_qr.py lib.so [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(sys.argv[1], "dqmc_")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
M = np.random.default_rng(1).standard_normal((256, 256)) * np.exp(np.random.default_rng(2).uniform(-6, 6, 256))[None, :]
for _ in range(5): lib.to_ldr(M)
out = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(reps): lib.to_ldr(M)
    out.append((time.perf_counter() - t0) / reps * 1e6)
print(f"{os.path.basename(sys.argv[1]):40s} to_ldr us/call {np.round(out, 1)}")
