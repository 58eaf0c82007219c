#!/usr/bin/env python3
"""Launch-free factorisation (BGP_PERSIST=1) against the multi-launch path: bit-identity of the log-likelihoods and wall
time per LML call for small batches.  Runs each setting in a child process (the switch is read at context creation)."""
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
for n, d, B in %r:
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * rng.randn(B, d + 2)
    if B > 2: H[1, 1:d+1] = 8.0; H[1, d+1] = -40.0   # one matrix that is not positive definite (flat kernel, no noise)
    v, st = ctx.lml(H, return_status=True)
    for _ in range(4): ctx.lml(H)
    ts = []
    for _ in range(%d):
        t0 = time.perf_counter(); v2 = ctx.lml(H); ts.append(time.perf_counter() - t0)
    out["%%d_%%d_%%d" %% (n, d, B)] = {"ms": float(np.median(ts) * 1e3), "lml": [float(x).hex() for x in v], "status": st.tolist(),
                                   "stable": bool(np.array_equal(v, v2))}
    ctx.close()
print("RESULT " + json.dumps(out))
"""

def run(env, shapes, reps):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", CHILD % (root, shapes, reps)], env=dict(os.environ, **env),
                         capture_output=True, text=True, timeout=300)
    if res.returncode != 0:
        print(res.stderr[-3000:])
        raise SystemExit(1)
    if res.stderr.strip():
        print("stderr:", res.stderr.strip()[-1500:])
    return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])

if __name__ == "__main__":
    shapes = [(256, 4, 8), (1024, 8, 32), (975, 8, 50), (1024, 8, 16), (2048, 16, 16), (2048, 16, 32), (2048, 16, 64), (4096, 32, 1),
              (4096, 32, 8), (512, 8, 50)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    a = run({"BGP_PERSIST": "0"}, shapes, 30)
    b = run({"BGP_PERSIST": "1", "BGP_PS_PAIR": "0", "BGP_PS_TIMEOUT_MS": os.environ.get("BGP_PS_TIMEOUT_MS", "500")}, shapes, 30)
    c = run({"BGP_PERSIST": "1", "BGP_PS_PAIR": "1", "BGP_PS_TIMEOUT_MS": os.environ.get("BGP_PS_TIMEOUT_MS", "500")}, shapes, 30)
    for k in a:
        same = a[k]["lml"] == b[k]["lml"] and a[k]["status"] == b[k]["status"]
        samec = a[k]["lml"] == c[k]["lml"] and a[k]["status"] == c[k]["status"]
        print(f"{k:>14s}: launches {a[k]['ms']:.3f} ms   launch-free {b[k]['ms']:.3f} ms (x{a[k]['ms'] / b[k]['ms']:.2f})   chain pairs "
              f"{c[k]['ms']:.3f} ms (x{a[k]['ms'] / c[k]['ms']:.2f})   bit-identical {same} / {samec}   stable {b[k]['stable']} / "
              f"{c[k]['stable']}   failed matrices {sum(1 for s in b[k]['status'] if s)}", flush=True)
