#!/usr/bin/env python3
"""Dumps the per-kernel statistics of a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite output) as CSV.
usage: rocpd_to_csv.py <results.db> <out.csv>"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], round(r[2], 3), round(r[3], 3), round(r[4], 3)])
print(len(rows), "kernels ->", sys.argv[2])
