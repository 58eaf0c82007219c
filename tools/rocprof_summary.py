#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (rocpd sqlite) or kernel_stats CSV into a small text table:
per-kernel calls, total / average / min / max duration.  Usage: rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels group by {name_col} order by 3 desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
    for name, calls, tot, avg, mn, mx in rows:
        short = name.split("(")[0][-60:]
        lines.append(f"{short:60s} {calls:7d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
