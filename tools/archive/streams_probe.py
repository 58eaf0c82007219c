#!/usr/bin/env python3
"""Wall time per LML batch call at BASELINE config C (n=2048, d=16, B=128) for 1..4 walker-group streams."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd  # noqa
from bayes_skopt_amd import _lib
n, d, B = 2048, 16, 128
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.01 * rng.randn(B, d + 2)
ref = None
for ns in (1, 2, 3, 4, 1, 2):
    ctx.set_streams(ns)
    for _ in range(3): out = ctx.lml(H)
    t0 = time.perf_counter()
    for _ in range(20): out = ctx.lml(H)
    dt = (time.perf_counter() - t0) / 20 * 1e3
    if ref is None: ref = out
    print(f"streams={ns}: {dt:.3f} ms/call  ({B / dt * 1e3:.0f} evals/s)  identical to 1-stream result: {np.array_equal(out, ref)}", flush=True)
