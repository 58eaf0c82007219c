#!/usr/bin/env python3
"""The MCMC loop of BASELINE config B (n = 1024, 64 walkers) and the launch-free comparison of bench.py with the launch-free
factorisation off / forced / automatic: does what tools/persist_probe.py measures per LML call arrive in the sampler?"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json
sys.path.insert(0, %r)
import bench
import bayes_skopt_amd as bask
from bayes_skopt_amd import _lib
r = bench.config_b(bask, 0, steps=150)
lf = bench.launch_free(_lib, 0)
print("RESULT " + json.dumps({"ms_per_half_step": r["ms_per_half_step"], "evals_per_s": r["evals_per_s"], "launch_free": lf}))
""" % ROOT

for tag, env in (("off", {"BGP_PERSIST": "0"}), ("on", {"BGP_PERSIST": "1"}), ("auto", {})):
    e = dict(os.environ)
    e.pop("BGP_PERSIST", None)
    e.update(env)
    res = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=600)
    if res.returncode != 0:
        print(tag, "FAILED", res.stderr[-2000:])
        continue
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    print(f"{tag:5s} config B: {d['ms_per_half_step']:.3f} ms per half-step, {d['evals_per_s']:.0f} evals/s;  "
          + "  ".join(f"{k}: {v['launches_ms']:.3f} / {v['launch_free_ms']:.3f}" for k, v in d["launch_free"].items()), flush=True)
    if res.stderr.strip():
        print("  stderr:", res.stderr.strip()[-400:])
