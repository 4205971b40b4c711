"""Where a panel of the panel-pivoted QR spends its time: runs dqmc_to_ldr once on a graded n x n matrix with the stamp build of
qr_panel.hip (scripts/stamp_build/libdqmc_hip_qpst.so: hipcc -DDQ_QP_STAMPS, s_memtime at every step; diagnostic, never used for
timing results) -- the kernel prints the ticks of panel k = 0.   usage: qp_stamps.py [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(os.environ.get("DQMC_LIB", os.path.join(ROOT, "scripts", "stamp_build", "libdqmc_hip_qpst.so")), "dqmc_")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(3)
M = rng.standard_normal((n, n)) * np.exp(rng.uniform(-6, 6, n))[None, :]
for _ in range(3):
    lib.to_ldr(M)
