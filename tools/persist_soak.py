#!/usr/bin/env python3
"""Soak of the launch-free factorisation: many calls on the shapes it is switched on for automatically; every result
compared with the first one (bitwise), timeouts counted (stderr).  usage: persist_soak.py [calls]"""
import os, sys, time
os.environ.setdefault("BGP_PERSIST", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayes_skopt_amd  # noqa
from bayes_skopt_amd import _lib
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
# (single chain workgroup: 2048 x 16, 2048 x 9, 4096 x 8, 3000 x 5; chain pairs: 4096 x 1 / 2, 2048 x 1 / 8, 1536 x 12, 1024 x 16, 640 x 3)
for n, d, B in ((2048, 16, 16), (4096, 32, 1), (2048, 16, 9), (4096, 32, 8), (2048, 16, 1), (3000, 8, 5), (4096, 32, 2), (2048, 16, 8),
                (1536, 12, 12), (1024, 8, 16), (640, 4, 3)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * rng.randn(B, d + 2)
    ref = ctx.lml(H)
    bad = 0
    t0 = time.perf_counter()
    for i in range(calls):
        v = ctx.lml(H + (0.0 if i % 2 else 1e-3))  # two alternating blocks
        if i % 2 and not np.array_equal(v, ref): bad += 1
    dt = (time.perf_counter() - t0) / calls * 1e3
    print(f"n={n} B={B}: {calls} calls, {dt:.3f} ms per call, mismatches {bad}", flush=True)
    ctx.close()
