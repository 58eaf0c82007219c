#!/usr/bin/env python3
"""Where do the factors of the launch-free and the multi-launch path differ?  usage: persist_lcmp.py n d B"""
import os, subprocess, sys, json
import numpy as np
CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
n, d, B = %d, %d, %d
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * rng.randn(B, d + 2)
v = ctx.lml(H)
Ls, zs = zip(*[ctx.debug_workspace(b) for b in range(B)])
np.savez(%r, v=v, L=np.array(Ls), z=np.array(zs))
"""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, d, B = (int(a) for a in sys.argv[1:4])
for tag, env in (("launch", {"BGP_PERSIST": "0"}), ("persist", {"BGP_PERSIST": "1", "BGP_PS_TIMEOUT_MS": "400"})):
    r = subprocess.run([sys.executable, "-c", CHILD % (root, n, d, B, "/tmp/lcmp_%s.npz" % tag)], env=dict(os.environ, **env), capture_output=True, text=True)
    if r.returncode or r.stderr.strip(): print(tag, r.stderr[-800:])
a, b = np.load("/tmp/lcmp_launch.npz"), np.load("/tmp/lcmp_persist.npz")
print("lml equal:", (a["v"] == b["v"]).tolist())
for m in range(B):
    La, Lb = np.tril(a["L"][m]), np.tril(b["L"][m])
    neq = np.argwhere(La != Lb)
    zneq = np.flatnonzero(a["z"][m] != b["z"][m])
    if len(neq) or len(zneq):
        blk = sorted(set((int(i) // 128, int(j) // 128) for i, j in neq))
        print("matrix", m, "L entries differing:", len(neq), "first:", neq[:4].tolist(), "blocks:", blk[:12], "max abs diff %.3e" % np.abs(La - Lb).max(),
              "| z rows differing:", len(zneq), zneq[:8].tolist())
for m in range(B):
    for tag, r in (("launch", a), ("persist", b)):
        L, z = r["L"][m], r["z"][m]
        host = -0.5 * float(z @ z) - float(np.log(np.diag(L)).sum()) - 0.5 * n * np.log(2 * np.pi)
        print(m, tag, "device lml %.17g  host-from-(L,z) %.17g  diff %.3e   z[n:] nonzero: %d  diag(L)[n:] != 1: %d" % (
            r["v"][m], host, r["v"][m] - host, int(np.count_nonzero(z[n:])), int(np.count_nonzero(np.diag(L)[n:] != 1.0))))
