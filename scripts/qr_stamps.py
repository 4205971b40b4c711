"""to_LDR of a graded 256 x 256 matrix on a stamp build of the column-owner QRCP (-DDQ_QR_STAMPS, scripts/stamp_build/libdqmc_hip_qr.so):
prints the per-phase cycle totals the kernel reports (diagnostic only)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stamp_build", sys.argv[1] if len(sys.argv) > 1 else "libdqmc_hip_qr.so"), "dqmc_")
M = np.random.default_rng(1).standard_normal((256, 256)) * np.exp(np.random.default_rng(2).uniform(-6, 6, 256))[None, :]
lib.to_ldr(M); lib.to_ldr(M)
