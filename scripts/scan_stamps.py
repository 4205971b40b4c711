"""Runs a few slices of the cfg-3 local update on the stamp build (scripts/scan_stamps.sh) and prints
the per-window cycle breakdown the kernel reports (diagnostic only; never used for timing results)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dqmc_amd import HubbardModel, CONFIGS
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stamp_build", "libdqmc_hip_stamps.so"), "dqmc_")
m = HubbardModel(**CONFIGS["cfg3"])
# thermalise with the production library first, so that the stamped slices see the acceptance rate of the benchmark
import dqmc_amd
e0 = m.engine(dqmc_amd.lib()); e0.set_fields(m.random_fields(3)); e0.init()
rng = np.random.default_rng(0)
for _ in range(3):
    e0.sweep_0_to_beta(*m.random_stream(rng)); e0.sweep_beta_to_0(*m.random_stream(rng))
fields = e0.get_fields(); del e0
e = m.engine(lib); e.set_fields(fields); e.init()
for l in range(3):
    e.wrap_forward(l)
    print("accepted:", e.local_update_slice(l, *m.random_stream(rng, 1)), flush=True)

import numpy as np
M = np.random.default_rng(1).standard_normal((256, 256)) * np.exp(np.random.default_rng(2).uniform(-6, 6, 256))[None, :]
print("to_ldr 256:"); lib.to_ldr(M); lib.to_ldr(M)
