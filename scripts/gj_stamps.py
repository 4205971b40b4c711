"""Where a step of the Gauss-Jordan panel spends its time: one inv(I + F1 F2) with the stamp build of lu_gj.hip
(scripts/stamp_build/libdqmc_hip_gjst.so: hipcc -DDQ_GJ_STAMPS, s_memtime between the phases of every step, summed over the panel;
diagnostic, the waits it inserts change the overlap -- never used for timing results).   usage: gj_stamps.py [n ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(os.environ.get("DQMC_LIB", os.path.join(ROOT, "scripts", "stamp_build", "libdqmc_hip_gjst.so")), "dqmc_")
for n in [int(a) for a in sys.argv[1:]] or [128, 256, 576]:
    rng = np.random.default_rng(3)
    F = []
    for _ in range(2):
        Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
        F.append((np.asfortranarray(Q), np.exp(np.sort(rng.uniform(-4, 4, n))[::-1]), np.asfortranarray(np.triu(rng.standard_normal((n, n)) * 0.2, 1) + np.eye(n))))
    for _ in range(2):
        lib.inv_I_plus_ldr_mul_ldr(F[0], F[1])
