#!/usr/bin/env python3
"""Per-kernel PMC summary from rocprofv3 --pmc rocpd databases.
Usage: rocprof_pmc_summary.py out.txt db1 [db2 ...]   (one counter set per db / pass)"""
import sqlite3
import sys


def main():
    out, dbs = sys.argv[1], sys.argv[2:]
    lines = []
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute(
            "select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name, counter_name"
        ).fetchall()
        lines.append(f"# {db.split('/')[-1]}")
        lines.append(f"{'kernel':44s} {'counter':34s} {'dispatches':>10s} {'sum':>16s} {'avg/dispatch':>16s} {'avg_us':>9s}")
        for name, cn, cnt, sm, av, dur in rows:
            if name.startswith("__amd"):
                continue
            short = name.split("(")[0].replace("void ", "")[-44:]
            lines.append(f"{short:44s} {cn:34s} {cnt:10d} {sm:16.1f} {av:16.2f} {dur/1e3:9.1f}")
        lines.append("")
    text = "\n".join(lines)
    print(text)
    open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
