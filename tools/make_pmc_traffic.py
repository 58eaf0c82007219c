#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from the rocprofv3 PMC passes of tools/profile_round.sh: HBM bytes per launch of the
trailing-update kernel class (FETCH_SIZE x 2 -- the gfx950 correction of MI355X_MICROARCH.md section HBM -- plus
WRITE_SIZE; both counters are reported in KiB... units of 1024 B by this rocprofv3) and its MFMA utilisation.
Usage: make_pmc_traffic.py out.json fetch.db write.db mfma.db kernel-substring ["what ran"]"""
import json
import sqlite3
import sys


def per_kernel(db, sub):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    out = {}
    for name, cn, cnt, sm in rows:
        if sub in name:
            c, s = out.get(cn, (0, 0.0))
            out[cn] = (c + cnt, s + sm)
    return out


def main():
    out, fdb, wdb, mdb, sub = sys.argv[1:6]
    what = sys.argv[6] if len(sys.argv) > 6 else "tools/profile_round.sh: bench.py --steps 2 --warmup 1 --no-extras (config C only)"
    f = per_kernel(fdb, sub)["FETCH_SIZE"]
    w = per_kernel(wdb, sub)["WRITE_SIZE"]
    m = per_kernel(mdb, sub)
    fetch_kb, write_kb = f[1] / f[0], w[1] / w[0]
    rec = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES in separate passes, "
                  "BGP_STREAMS=1; " + what,
        "kernel": sub,
        "launches_averaged": int(f[0]),
        "fetch_size_kb_per_launch_raw": fetch_kb,
        "fetch_size_correction": 2.0,
        "fetch_size_correction_note": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a "
                                      "wide coalesced streaming read (16 B / lane, global_load and LDS-DMA alike)",
        "write_size_kb_per_launch_raw": write_kb,
        "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    }
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # 4 SIMDs x 256 CUs = 1024 MFMA pipes; GRBM_GUI_ACTIVE is summed over the 8 XCDs -> pipes per XCD-cycle = 128
        rec["mfma_util_percent"] = 100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (m["GRBM_GUI_ACTIVE"][1] * 128.0)
        rec["mfma_util_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)"
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
