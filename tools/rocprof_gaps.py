#!/usr/bin/env python3
"""Timeline of the last N kernels of a rocprofv3 results .db: name, duration, gap to the previous kernel's end.
Usage: rocprof_gaps.py <results.db> [N]"""
import sqlite3
import sys

db = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
rows = rows[-N:]
prev_end = None
tot_k = tot_gap = 0
for name, s, e in rows:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{name.split('(')[0][-40:]:40s} dur {(e - s) / 1e3:8.2f} us   gap {gap:8.2f} us")
    tot_k += e - s
    if prev_end is not None and gap < 200:
        tot_gap += max(s - prev_end, 0)
    prev_end = e
print(f"sum kernel {tot_k / 1e3:.1f} us, sum gaps(<200us) {tot_gap / 1e3:.1f} us, span {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us")
