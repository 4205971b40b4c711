"""Stress of several engines sweeping at once on one device (diagnostic): two identical worlds of W replicas (cfg-4 size), world A
sweeps with its W host threads running concurrently, world B one replica after the other; after every sweep the HS fields must be
identical and G equal to rounding.  usage: pt_stress.py [iterations] [W] [concurrent: 1 | 0]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from pt_twin import HostPT, ini_text, load_host
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
conc = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
L, U, nt, ns = 16, 8.0, 200, 10; n = L * L
betas = [8.0 - 0.4 * r for r in range(W)]; seeds = [1000 + r for r in range(W)]
h = load_host(); ini = ini_text(L, U, nt, ns)
A = HostPT(h, ini, betas, seeds); B = HostPT(h, ini, betas, seeds)
bad = 0; t0 = time.time()
for it in range(iters):
    A.sweeps(1, conc); B.sweeps(1, False)
    for r in range(W):
        a, b = A.get(r, n, nt), B.get(r, n, nt)
        nd = int((a["fields"] != b["fields"]).sum()); dG = float(np.abs(a["G"] - b["G"]).max())
        if nd or dG > 1e-7 * max(1.0, float(np.abs(b["G"]).max())):
            bad += 1
            sl = np.nonzero((a["fields"] != b["fields"]).any(axis=1))[0]
            per = (a["fields"] != b["fields"]).sum(axis=1)
            print(f"iteration {it} replica {r}: {nd} field entries differ in {len(sl)} slices, lowest {sl[:6]}, highest {sl[-6:]}, per slice (highest first) {per[sl][::-1][:12]}, "
                  f"max|dG| {dG:.3e}, max|G| {np.abs(b['G']).max():.2e}, max wrap err A {A.max_err(r):.3e} B {B.max_err(r):.3e}, accepted A/B {a['accepted']}/{b['accepted']}", flush=True)
            A.set_fields(r, b["fields"])            # resynchronise and go on
    if it % 50 == 49:
        print(f"{it + 1} iterations, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("done:", iters, "iterations,", bad, "mismatches")
A.close(); B.close()
