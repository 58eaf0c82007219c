#!/usr/bin/env python3
"""Per-(kernel, grid size) durations from a rocprofv3 results .db: rocprof_by_grid.py <results.db> [name-filter]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gcol = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")]
gcol = gcol[0] if gcol else None
if gcol is None:
    print("columns:", cols)
    sys.exit(0)
rows = cur.execute(f"select name, {gcol}, count(*), avg(end-start), min(end-start) from kernels group by name, {gcol} "
                   f"order by name, {gcol} desc").fetchall()
for name, g, c, avg, mn in rows:
    short = name.split("(")[0][-40:]
    if flt in short:
        print(f"{short:40s} grid {g:9d} calls {c:5d} avg {avg/1e3:9.2f} us  min {mn/1e3:9.2f} us")
