"""Runs a few slices of the cfg-3 local update on a stamp build of the sub-matrix slice kernel (-DDQ_SM_STAMPS, library at
scripts/stamp_build/libdqmc_hip.so) and prints the cycle breakdown the kernel reports (diagnostic only)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dqmc_amd import fixtures
from dqmc_amd.abi import DqmcLib
lib = DqmcLib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stamp_build", sys.argv[1] if len(sys.argv) > 1 else "libdqmc_hip.so"), "dqmc_")
z, m, streams = fixtures.load("cfg3_therm")
e = m.engine(lib); e.set_fields(z["fields"]); e.init()
for l in range(4):
    e.wrap_forward(l)
    print("accepted:", e.local_update_slice(l, streams[0][0][l], streams[0][1][l], streams[0][2][l]), flush=True)
