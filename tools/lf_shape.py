#!/usr/bin/env python3
"""ONE shape of the launch-free factorisation, alone in a process, for the rocprofv3 counter passes of tools/profile_lf_pmc.sh:
  lf_shape.py lml n d B reps     `reps` forced launch-free LML calls of B matrices of order n
  lf_shape.py cov n d m reps     `reps` sample_y calls over m candidates: ONE (m padded to 128)^2 covariance factorisation each
Prints one JSON line with the wall time per call."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayes_skopt_amd  # noqa: E402,F401
from bayes_skopt_amd import _lib  # noqa: E402

kind, n, d, B, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d))
y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
y = (y - y.mean()) / y.std()
h = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
if kind == "lml":
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
    ctx.set_persist(1)
    H = h + 0.05 * rng.randn(B, d + 2)
    for _ in range(3):
        ctx.lml(H)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.lml(H)
    ms = (time.perf_counter() - t0) / reps * 1e3
else:
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=2)
    ctx.set_persist(1)
    ctx.posterior(h[None, :])
    hk = h.copy()
    hk[-1] = -np.inf
    Xq = rng.uniform(size=(B, d))
    z = rng.randn(1, B)
    ctx.sample_y(0, hk, Xq, z, jitter=1e-8)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.sample_y(0, hk, Xq, z, jitter=1e-8)
    ms = (time.perf_counter() - t0) / reps * 1e3
st = ctx.persist_stats()
ctx.close()
print(json.dumps({"kind": kind, "n": n, "d": d, "B_or_m": B, "ms_per_call": ms, "launch_free_calls": st["calls"], "timeouts": st["timeouts"]}))
