"""Small workload for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE need separate passes and serialise
every dispatch, so the full bench is too long): cfg-3 init + a few slices of wrap + local update."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dqmc_amd
lib = dqmc_amd.lib()
m = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS["cfg3"])
e = m.engine(lib); e.set_fields(m.random_fields(12345)); e.init()
rng = np.random.default_rng(0)
nslices = int(sys.argv[1]) if len(sys.argv) > 1 else 4
acc = 0
for l in range(nslices):
    e.wrap_forward(l)
    acc += e.local_update_slice(l, *m.random_stream(rng, 1))
print("slices", nslices, "accepted", acc, flush=True)
