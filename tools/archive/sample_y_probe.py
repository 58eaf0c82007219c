#!/usr/bin/env python3
"""bgp_sample_y over 10 000 candidates at n ~ 1000 (the Thompson draw of a config-E PVRS tell: one 10 112 x 10 112 covariance
Cholesky, 79 block columns): wall time and bit-identity, launch schedule vs launch-free factorisation (BGP_PERSIST)."""
import json, os, subprocess, sys
CHILD = r"""
import sys, json, time
sys.path.insert(0, %r)
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
rng = np.random.RandomState(0)
n, d, m = 1000, 8, 10000
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=8)
h = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
ctx.posterior(h[None, :])
hk = h.copy(); hk[-1] = -np.inf
Xq = rng.uniform(size=(m, d)); z = rng.randn(5, m)
out = ctx.sample_y(0, hk, Xq, z, jitter=1e-8)
ts = []
for _ in range(6):
    t0 = time.perf_counter(); out = ctx.sample_y(0, hk, Xq, z, jitter=1e-8); ts.append((time.perf_counter() - t0) * 1e3)
print("RESULT " + json.dumps({"ms": float(np.median(ts)), "sum": float(out.sum()).hex(), "first": float(out[0, 0]).hex()}))
"""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for tag, env in (("launches", {"BGP_PERSIST": "0"}), ("launch-free", {"BGP_PERSIST": "1", "BGP_PS_TIMEOUT_MS": "2000"})):
    r = subprocess.run([sys.executable, "-c", CHILD % root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    if r.returncode: print(tag, r.stderr[-1500:]); continue
    res[tag] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    print(tag, res[tag])
if len(res) == 2:
    print("bit-identical draws:", res["launches"]["sum"] == res["launch-free"]["sum"] and res["launches"]["first"] == res["launch-free"]["first"])
