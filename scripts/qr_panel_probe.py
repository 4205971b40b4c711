"""Panel-pivoted to_LDR on the device against the column-pivoted kernels and the numpy statement of the same algorithm.
usage: qr_panel_probe.py [n ...]     (run it twice, with DQMC_QR_PANEL=0 and without, for the A/B timing)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dqmc_amd.abi import DqmcLib
from oracle import panel_qr as pq

lib = DqmcLib(os.environ.get("DQMC_LIB", os.path.join(ROOT, "dqmc_amd", "libdqmc_hip.so")), "dqmc_")
sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 576]
mode = "column-pivoted" if os.environ.get("DQMC_QR_PANEL") == "0" else "panel"
for n in sizes:
    rng = np.random.default_rng(100 + n)
    for kind in ("col-graded", "row-graded", "both"):
        X = rng.standard_normal((n, n)); s1 = np.exp(rng.uniform(-12, 12, n)); s2 = np.exp(rng.uniform(-12, 12, n))
        M = X * s1[None, :] if kind == "col-graded" else (s1[:, None] * X if kind == "row-graded" else s1[:, None] * X * s2[None, :])
        L, d, R = lib.to_ldr(M)
        orth = np.abs(L.T @ L - np.eye(n)).max()
        rec = np.abs((L * d[None, :]) @ R - M).max() / np.abs(M).max()
        # column-wise relative reconstruction (a graded matrix hides its small columns behind max|M|)
        recc = (np.abs((L * d[None, :]) @ R - M).max(axis=0) / np.abs(M).max(axis=0)).max()
        suf = np.maximum.accumulate(d[::-1])[::-1]
        grade = (suf[1:] / d[:-1]).max()
        rmax = np.abs(R).max()
        Qn, R0n, Pn = pq.qr_sketch(M, 16, 16, local_pivot=False, sign=True)
        dn = np.abs(np.diag(R0n))
        same = np.allclose(d, dn, rtol=1e-6)
        print(f"{mode:15s} n={n:4d} {kind:10s} |LtL-I| {orth:.1e}  rec {rec:.1e} (col-wise {recc:.1e})  max|R| {rmax:.2f}  grading {grade:.2f}  d == numpy sketch-QR: {same}", flush=True)
    M = rng.standard_normal((n, n)) * np.exp(rng.uniform(-6, 6, n))[None, :]
    for _ in range(3): lib.to_ldr(M)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps): lib.to_ldr(M)
    print(f"{mode:15s} n={n:4d} to_ldr wall {1e6 * (time.perf_counter() - t0) / reps:.0f} us per call (incl. upload / download)", flush=True)
