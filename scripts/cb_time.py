"""Time the checkerboard pair kernel in situ: 200 wrap pairs back to back on an engine's stream -> microseconds per launch
(2 launches per wrap), beside the dense-GEMM wraps of the same engine size.  usage: python scripts/cb_time.py [cfg ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dqmc_amd

lib = dqmc_amd.lib()
for cfg in (sys.argv[1:] or ["cfg3", "cfg5"]):
    m = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS[cfg])
    for cb in (False, True):
        e = m.engine(lib)
        if cb:
            e.set_checkerboard(*m.checkerboard())
        e.set_fields(m.random_fields(1)); e.set_G(np.eye(m.n))
        for _ in range(10):
            e.wrap_forward(0); e.wrap_backward(0)
        e.sync(); t0 = time.perf_counter()
        for _ in range(200):
            e.wrap_forward(0); e.wrap_backward(0)
        e.sync(); dt = time.perf_counter() - t0
        print(f"{cfg} n={m.n} {'checkerboard' if cb else 'dense GEMM  '} {1e6 * dt / 800:.2f} us per launch", flush=True)
        e.close()
