#!/usr/bin/env python3
"""Per-kernel averages of the counters of a rocprofv3 `--pmc ... --kernel-trace` run (rocpd sqlite output) as CSV.
usage: rocpd_pmc_to_csv.py <results.db> <out.csv>"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = next((t for t in tables if t == "counters_collection"), None)
if view is None:
    print("tables/views:", tables); raise SystemExit("no counters_collection view in this database")
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
kcol = "kernel_name" if "kernel_name" in cols else next(c for c in cols if "kernel" in c and "name" in c)
rows = list(db.execute(f"select {kcol}, counter_name, count(*), avg(value) from {view} group by {kcol}, counter_name order by {kcol}"))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "counter", "dispatches", "avg_value_per_dispatch"])
    for r in rows:
        w.writerow([r[0][:80], r[1], r[2], round(r[3], 3)])
print(len(rows), "rows ->", sys.argv[2], "| columns:", cols)
