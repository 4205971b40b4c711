#!/usr/bin/env python3
"""The kernels of ONE stabilisation in launch order (name, duration, gap to the previous kernel) from a rocprofv3 --kernel-trace rocpd
database: the stretch between two consecutive slice-kernel launches that holds the median number of kernels among the long ones (a steady-state stabilisation).
usage: rocpd_sequence.py <results.db> [slice kernel name fragment]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); frag = sys.argv[2] if len(sys.argv) > 2 else "slice_kernel"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else next(c for c in cols if "name" in c)
rows = list(db.execute(f"select {name}, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if frag in r[0]]
if len(idx) < 2: raise SystemExit("fewer than two launches of " + frag)
# stretches with more than 20 kernels are stabilisations; the MEDIAN one by kernel count is a steady-state stabilisation (the longest is the
# first after an initialisation, whose stack still holds product R factors and goes through the blocked LU)
cands = sorted((idx[j + 1] - idx[j], j) for j in range(len(idx) - 1) if idx[j + 1] - idx[j] > 20)
if not cands: raise SystemExit("no stretch with more than 20 kernels between two launches of " + frag)
best = cands[len(cands) // 2][1]
a, b = idx[best], idx[best + 1]
prev_end = rows[a][2]; tot = 0.0; gaps = 0.0
print(f"{b - a - 1} kernels between two launches of {frag}: wall {(rows[b][1] - rows[a][2]) / 1e3:.1f} us")
for r in rows[a + 1:b]:
    d = (r[2] - r[1]) / 1e3; g = (r[1] - prev_end) / 1e3; tot += d; gaps += g; prev_end = r[2]
    print(f"  {d:8.1f} us  gap {g:6.1f}  {r[0][:100]}")
print(f"sum of kernel durations {tot:.1f} us, of gaps {gaps:.1f} us")
