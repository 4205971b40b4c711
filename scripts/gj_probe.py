"""Gauss-Jordan panel A/B: inv(I + F1 F2) through the stateless ABI at several sizes; prints a digest of the result bits and the error
against numpy, so two builds / switches can be compared bit for bit.  usage: gj_probe.py [n ...]"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dqmc_amd.abi import DqmcLib

lib = DqmcLib(os.environ.get("DQMC_LIB", os.path.join(ROOT, "dqmc_amd", "libdqmc_hip.so")), "dqmc_")
for n in [int(a) for a in sys.argv[1:]] or [72, 100, 256, 320, 576, 600]:
    rng = np.random.default_rng(7 + n)
    F = []
    for _ in range(2):
        Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
        d = np.exp(np.sort(rng.uniform(-8, 8, n))[::-1])
        R = np.triu(rng.standard_normal((n, n)) * 0.2, 1) + np.eye(n)
        F.append((np.asfortranarray(Q), d, np.asfortranarray(R)))
    G = lib.inv_I_plus_ldr_mul_ldr(F[0], F[1])
    M = (F[0][0] * F[0][1]) @ F[0][2] @ (F[1][0] * F[1][1]) @ F[1][2]
    ref = np.linalg.inv(np.eye(n) + M)
    t0 = time.perf_counter()
    for _ in range(20): lib.inv_I_plus_ldr_mul_ldr(F[0], F[1])
    dt = (time.perf_counter() - t0) / 20
    print(f"n={n:4d} digest {hashlib.sha1(np.ascontiguousarray(G).tobytes()).hexdigest()[:16]}  max|G - numpy| {np.abs(G - ref).max():.2e}  wall {1e6 * dt:.0f} us", flush=True)
