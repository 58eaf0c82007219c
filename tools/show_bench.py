#!/usr/bin/env python3
"""Print the parts of a bench.py line that the round's review items are about."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "e2e", d["end_to_end"]["frac"])
for key in ("config_A", "config_B"):
    c = d.get(key, {})
    print(key, {k: v for k, v in c.items() if k != "cpu_baseline"}, "cpu:", c.get("cpu_baseline", {}).get("value"))
print("E", d.get("config_E"))
for k in d.get("roofline_kernels", []):
    print(k)
print(d["roofline"]["by_launch_kind"])
print("shard", d.get("shard_ms"))
print("lf", d.get("launch_free"))
print("kernel_ms", d["kernel_ms_per_half_step"], "fit+sample", d.get("fit_plus_sample_ms"))
