#!/usr/bin/env python3
"""How far apart are the launch-free and the multi-launch log-likelihoods (should be 0)?  usage: persist_diff.py n,d,B ..."""
import json, os, subprocess, sys
CHILD = r"""
import sys, json
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
    vs = [ctx.lml(H).tolist() for _ in range(6)]
    out["%%d_%%d_%%d" %% (n, d, B)] = vs
    ctx.close()
print("RESULT " + json.dumps(out))
"""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
res = {}
for tag, env in (("launch", {"BGP_PERSIST": "0"}), ("persist", {"BGP_PERSIST": "1", "BGP_PS_TIMEOUT_MS": "400"})):
    r = subprocess.run([sys.executable, "-c", CHILD % (root, shapes)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    if r.stderr.strip(): print(tag, "stderr:", r.stderr.strip()[-600:])
    res[tag] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
import numpy as np
for k in res["launch"]:
    a = np.array(res["launch"][k]); b = np.array(res["persist"][k])
    rel = np.abs(b - a[0]) / np.abs(a[0])
    print(k, "launch self-consistent", bool((a == a[0]).all()), " persist calls: max rel diff per call", ["%.2e" % v for v in rel.max(axis=1)],
          " matrices differing in call 0:", int((rel[0] > 0).sum()), "of", a.shape[1])
