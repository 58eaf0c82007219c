#!/usr/bin/env python3
"""BGP_FUSED_TRSM=0 / 1 (the group's pending panels applied by the panel solve instead of look-ahead column launches): the same
bits?  wall time per LML call at a few shapes (launch schedule forced).  usage: fused_trsm_probe.py"""
import json
import os
import subprocess
import sys

CHILD = r"""
import os, sys, time, json
sys.path.insert(0, %r)
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
out = {}
for n, d, B in ((2048, 16, 128), (2048, 16, 32), (1024, 8, 64), (4096, 32, 8), (1536, 8, 50), (1000, 5, 20)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
    ctx.set_persist(0)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * rng.randn(B, d + 2)
    if B > 2: H[1, 1:d+1] = 8.0; H[1, d+1] = -40.0
    v, st = ctx.lml(H, return_status=True)
    for _ in range(3): ctx.lml(H)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); ctx.lml(H); ts.append(time.perf_counter() - t0)
    L, z = ctx.debug_workspace(0)
    out["%%d_%%d_%%d" %% (n, d, B)] = {"ms": float(np.median(ts) * 1e3), "lml": [float(x).hex() for x in v], "status": st.tolist(),
                                   "L": float(np.tril(L).sum()).hex()}
    ctx.close()
print("RESULT " + json.dumps(out))
"""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for f in ("0", os.environ.get("FT", "1")):
    r = subprocess.run([sys.executable, "-c", CHILD % root], env=dict(os.environ, BGP_FUSED_TRSM=f), capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(r.stderr[-3000:]); raise SystemExit(1)
    res[f] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
for k in res["0"]:
    a, b = res["0"][k], res[os.environ.get("FT", "1")][k]
    print(f"{k:>14s}: look-ahead columns {a['ms']:.3f} ms   fused into the panel solve {b['ms']:.3f} ms (x{a['ms'] / b['ms']:.3f})   "
          f"bit-identical {a['lml'] == b['lml'] and a['status'] == b['status'] and a['L'] == b['L']}")
