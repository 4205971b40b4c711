"""Runs a few slices of the cfg-3 local update on the stamp build (scripts/scan_stamps.sh) and prints
the per-window cycle breakdown the kernel reports (diagnostic only; never used for timing results)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dqmc_amd import HubbardModel, CONFIGS
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stamp_build", "libdqmc_hip_stamps.so"), "dqmc_")
m = HubbardModel(**CONFIGS["cfg3"])
e = m.engine(lib); e.set_fields(m.random_fields(3)); e.init()
rng = np.random.default_rng(0)
for l in range(3):
    e.wrap_forward(l)
    print("accepted:", e.local_update_slice(l, *m.random_stream(rng, 1)), flush=True)

import numpy as np
M = np.random.default_rng(1).standard_normal((256, 256)) * np.exp(np.random.default_rng(2).uniform(-6, 6, 256))[None, :]
print("to_ldr 256:"); lib.to_ldr(M); lib.to_ldr(M)
