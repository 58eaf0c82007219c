#!/usr/bin/env python3
"""Per-call breakdown of the LML batch at BASELINE config B (n=1024, d=8, 32 proposals per half-step)
and other small-batch shapes: kernel classes (HIP events), device total, host wall per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd  # noqa
from bayes_skopt_amd import _lib

def run(n, d, B, reps=50):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.01 * rng.randn(B, d + 2)
    for _ in range(5): ctx.lml(H)
    t0 = time.perf_counter()
    for _ in range(reps): ctx.lml(H)
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.set_timing(True)
    acc = {}
    for _ in range(10):
        ctx.lml(H); tm = ctx.last_timing()
        for k in ("kbuild", "potrf", "trsm", "syrk"):
            acc[k] = acc.get(k, 0.0) + tm[k]["ms"] / 10
        acc["dev"] = acc.get("dev", 0.0) + tm["device_total_ms"] / 10
    ctx.set_timing(False)
    flops = B * (n ** 3 / 3 + 2 * n * n + n * (n - 1) / 2 * (3 * d + 14))
    print(f"n={n} d={d} B={B}: wall {wall:.3f} ms/call ({B / wall * 1e3:.0f} evals/s, {flops / wall / 1e9:.1f} TF)  "
          + "  ".join(f"{k} {v:.3f}" for k, v in acc.items()), flush=True)
    ctx.close()

for n, d, B in ((1024, 8, 32), (1024, 8, 50), (512, 8, 50), (256, 8, 50), (128, 2, 50), (2048, 16, 128), (2048, 16, 16)):
    run(n, d, B)
