#!/usr/bin/env python3
"""Per-kernel statistics (calls, total / average / min / max duration, share) of a rocprofv3 --kernel-trace run whose output
is a rocpd sqlite database (rocprofv3 of ROCm 7 writes <name>_results.db), as CSV -- the table `--stats` used to print.
usage: rocpd_kernel_stats.py <results.db> <out.csv>"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else next(c for c in cols if "name" in c)
rows = list(db.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by {name} order by 3 desc"))
total = sum(r[2] for r in rows) or 1
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0][:120], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / total, 3), r[4], r[5]])
for r in rows[:16]:
    print(f"{100.0 * r[2] / total:6.2f}%  {r[1]:7d} x {r[3] / 1e3:10.1f} us  {r[0][:90]}")
