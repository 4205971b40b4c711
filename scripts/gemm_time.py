"""Time the single-matrix GEMM in situ (wraps back to back on an engine's stream) and check it against numpy.
usage: python scripts/gemm_time.py [L | L1xL2 ...]   (lattice lengths; default 16 18 20 24 32)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dqmc_amd

lib = dqmc_amd.lib()
rng = np.random.default_rng(1)
for spec in (sys.argv[1:] or ["16", "18", "20", "24", "32"]):
    L1, L2 = (int(x) for x in spec.split("x")) if "x" in spec else (int(spec), int(spec))
    n = L1 * L2
    A = rng.standard_normal((n, n)); B = rng.standard_normal((n, n))
    for ta in (False, True):
        got = lib.gemm(A, B, transA=ta); want = (A.T if ta else A) @ B
        assert np.abs(got - want).max() < 1e-10 * np.abs(want).max(), (n, ta)
    m = dqmc_amd.HubbardModel(L1, L2, 4.0, 2.0, 20)
    e = m.engine(lib); e.set_fields(m.random_fields(1)); e.set_G(np.eye(n))
    for _ in range(5):
        e.wrap_forward(0); e.wrap_backward(0)
    e.sync(); t0 = time.perf_counter()
    for _ in range(100):
        e.wrap_forward(0); e.wrap_backward(0)
    e.sync(); dt = time.perf_counter() - t0
    print(f"n={n}: {1e6 * dt / 400:.2f} us per GEMM launch ({2.0 * n ** 3 / (dt / 400) / 1e12:.1f} TFLOP/s)", flush=True)
    e.close()
