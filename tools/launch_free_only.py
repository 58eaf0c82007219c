#!/usr/bin/env python3
"""The launch-free factorisation ALONE, in one process, for rocprofv3 (tools/profile_launch_free.sh): config B's batch
(n = 1024 x 32), the N = 8 shard of config C (n = 2048 x 16) and one n = 4096 matrix, `reps` calls each on the launch
schedule and then launch-free.  Prints one JSON line with the wall times."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayes_skopt_amd  # noqa: E402,F401
from bayes_skopt_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out = {}
for n, d, B in ((1024, 8, 32), (2048, 16, 16), (4096, 32, 1)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * rng.randn(B, d + 2)
    rec = {}
    vals = {}
    for tag, mode in (("launches", 0), ("launch_free", 1)):
        ctx.set_persist(mode)
        for _ in range(3):
            vals[tag] = ctx.lml(H)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.lml(H)
        rec[tag + "_ms"] = (time.perf_counter() - t0) / reps * 1e3
    rec["bit_identical"] = bool(np.array_equal(vals["launches"], vals["launch_free"]))
    out[f"n{n}_B{B}"] = rec
    ctx.close()
print(json.dumps(out))
