"""Runs a few slices of the local update on a stamp build (-DDQ_SM_STAMPS for update_sm.hip, -DDQ_SCAN_STAMPS for update.hip; library
under scripts/stamp_build/) and prints the cycle breakdown the kernel reports (diagnostic only).
usage: sm_stamps.py <library file> [config]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dqmc_amd
from dqmc_amd import fixtures
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stamp_build", sys.argv[1]), "dqmc_")
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
if f"{cfg}_therm" in fixtures.NAMES:
    z, m, streams = fixtures.load(f"{cfg}_therm"); fields = z["fields"]; st = streams[0]
else:
    m = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS[cfg]); fields = m.random_fields(12345); st = m.random_stream(np.random.default_rng(0))
e = m.engine(lib); e.set_fields(fields); e.init()
for l in range(3):
    e.wrap_forward(l)
    print("accepted:", e.local_update_slice(l, st[0][l], st[1][l], st[2][l]), flush=True)
