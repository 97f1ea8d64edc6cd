#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 results database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes NAME_results.db):
kernel, grid size, calls, average / min / max duration in microseconds, total in milliseconds.  Usage: rocprof_db_stats.py FILE.db [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
q = (f"select s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z, count(*), avg(d.end - d.start), min(d.end - d.start), "
     f"max(d.end - d.start), sum(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by 1, 2 order by 7 desc")
print(f"{'kernel':48s} {'grid':>9s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'total ms':>9s}")
for name, grid, n, avg, mn, mx, tot in db.execute(q).fetchall()[:top]:
    print(f"{name[:48]:48s} {grid:9d} {n:6d} {avg / 1e3:9.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f} {tot / 1e6:9.2f}")
