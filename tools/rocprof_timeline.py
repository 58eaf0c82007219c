#!/usr/bin/env python3
"""Back-to-back timeline of a rocprofv3 kernel trace: per kernel name the mean duration and the mean idle gap in front of it
(start minus the previous kernel's end on the device), for launch-bound loops such as the device-resident sampler's half-step.
usage: rocprof_timeline.py <results.db> [skip-first-N]"""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = con.execute("select name, start, end from kernels order by start").fetchall()[skip:]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for name, s, e in rows:
    k = name.split("(")[0][-44:]
    dur[k] += e - s
    cnt[k] += 1
    if prev_end is not None:
        gap[k] += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
span = rows[-1][2] - rows[0][1]
print(f"{len(rows)} kernels over {span/1e3:.1f} us")
for k in sorted(dur, key=lambda k: -dur[k]):
    print(f"{k:44s} calls {cnt[k]:6d}  mean {dur[k]/cnt[k]/1e3:9.2f} us  idle in front {gap[k]/cnt[k]/1e3:7.2f} us")
print(f"busy {sum(dur.values())/span*100:.1f} % of the span, idle {sum(gap.values())/1e3:.1f} us in all")
