#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace results .db:
rocprof_gaps.py <results.db> [last-N-kernels]  ->  span, busy (union of kernel intervals), per-kernel-name gap BEFORE it."""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
span = rows[-1][2] - rows[0][1]
busy, cur_end = 0, rows[0][1]
gap_by = defaultdict(lambda: [0, 0.0])
dur_by = defaultdict(lambda: [0, 0.0])
for name, s, e in rows:
    short = name.split("(")[0].split("<")[0][-28:]
    if s > cur_end:
        g = gap_by[short]
        g[0] += 1
        g[1] += s - cur_end
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
    d = dur_by[short]
    d[0] += 1
    d[1] += e - s
print(f"kernels {len(rows)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us  idle {(span-busy)/1e3:.1f} us")
for k in sorted(dur_by, key=lambda k: -dur_by[k][1]):
    g = gap_by.get(k, [0, 0.0])
    print(f"{k:30s} calls {dur_by[k][0]:6d}  time {dur_by[k][1]/1e3:10.1f} us  gap before: {g[1]/1e3:9.1f} us over {g[0]} "
          f"({(g[1]/g[0]/1e3 if g[0] else 0):.2f} us each)")
