#!/usr/bin/env python3
"""Concurrency between kernels of different HIP streams in a rocprofv3 --kernel-trace results .db:
rocprof_overlap.py <results.db> [last-N-kernels] [timeline-rows]
For every kernel name: calls, total time, and the part of that time during which at least one OTHER kernel was running
(any queue / stream); then a timeline excerpt (start, end relative to the first row, queue / stream, name)."""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("stream_id", "queue_id", "queue", "stream") if c in cols), None)
sel = "name, start, end" + (", %s" % qcol if qcol else ", 0")
rows = con.execute(f"select {sel} from kernels order by start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
short = lambda nm: nm.split("(")[0].split("<")[0][-26:]
# overlap of each kernel with the union of all others (sweep; n is a few thousand: O(n * window) is fine)
tot = defaultdict(lambda: [0, 0.0, 0.0])
for i, (nm, s, e, q) in enumerate(rows):
    ov = []
    j = i - 1
    while j >= 0 and rows[j][1] > s - 50_000_000:  # look back 50 ms
        if rows[j][2] > s:
            ov.append((max(s, rows[j][1]), min(e, rows[j][2])))
        j -= 1
    j = i + 1
    while j < len(rows) and rows[j][1] < e:
        ov.append((rows[j][1], min(e, rows[j][2])))
        j += 1
    ov.sort()
    cov, cur = 0, s
    for a, b in ov:
        if b > cur:
            cov += b - max(a, cur)
            cur = b
    t = tot[short(nm)]
    t[0] += 1
    t[1] += e - s
    t[2] += cov
print("columns:", cols)
print(f"{'kernel':28s} {'calls':>6s} {'total_ms':>10s} {'concurrent_ms':>14s} {'frac':>6s}")
for k in sorted(tot, key=lambda k: -tot[k][1]):
    c, a, b = tot[k]
    print(f"{k:28s} {c:6d} {a/1e6:10.3f} {b/1e6:14.3f} {b/a if a else 0:6.2f}")
m = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = rows[len(rows) // 2][1]
print("\ntimeline excerpt (us from its first row):")
for nm, s, e, q in rows[len(rows) // 2: len(rows) // 2 + m]:
    print(f"  {(s - t0)/1e3:10.1f} {(e - t0)/1e3:10.1f}  dur {(e - s)/1e3:8.1f}  q {q}  {short(nm)}")
