import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, dqmc_amd
from dqmc_amd import fixtures
z, m, st = fixtures.load("cfg3_therm")
e = m.engine(dqmc_amd.lib()); e.set_fields(z["fields"]); e.init(); e.sync()
t0 = time.perf_counter()
for _ in range(10): e.init()
e.sync(); print("init ms", 1e3 * (time.perf_counter() - t0) / 10, "dG vs golden", np.abs(e.get_G() - z["G0"]).max())
