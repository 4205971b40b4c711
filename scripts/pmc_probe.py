"""Small workload for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE need separate passes and serialise every dispatch, so
the full bench is too long): init on the thermalised cfg-3 fixture (the state bench.py starts from) + a few slices of wrap +
local update with the fixture's random stream, i.e. the bench's acceptance (~100 accepted flips per slice).
usage: pmc_probe.py [n_slices] [config]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dqmc_amd
from dqmc_amd import fixtures
lib = dqmc_amd.lib()
nslices = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
if f"{cfg}_therm" in fixtures.NAMES:
    z, m, streams = fixtures.load(f"{cfg}_therm"); fields = z["fields"]; st = streams[0]
else:
    m = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS[cfg]); fields = m.random_fields(12345); st = m.random_stream(np.random.default_rng(0))
e = m.engine(lib); e.set_fields(fields); e.init()
acc = 0
for l in range(nslices):
    e.wrap_forward(l)
    acc += e.local_update_slice(l, st[0][l], st[1][l], st[2][l])
print("slices", nslices, "accepted", acc, "per slice", acc / nslices, flush=True)
