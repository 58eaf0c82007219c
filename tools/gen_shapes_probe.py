#!/usr/bin/env python3
"""Launch schedule with the Gram blocks generated inside the first panel group's updates (BGP_SYRK_GEN=1) against the Gram kernel
in front of the factorisation (=0), by shape: ms per LML batch call, alternating, best of a few."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from bayes_skopt_amd import _lib  # noqa: E402

shapes = [(1024, 8, 64), (1024, 8, 128), (1536, 16, 64), (1536, 16, 128), (2048, 16, 32), (2048, 16, 64), (2048, 16, 128),
          (3072, 16, 16), (4096, 32, 8), (4096, 32, 16), (640, 4, 256)]
for n, d, B in shapes:
    rng = np.random.RandomState(n + B)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * rng.randn(B, d + 2)
    ctx = _lib.Context(X, y, 1e-10, max_batch=B)
    ctx.set_persist(0)
    best = {"0": 1e9, "1": 1e9}
    out = {}
    for rep in range(4):
        for mode in ("0", "1"):
            os.environ["BGP_SYRK_GEN"] = mode
            ctx.lml(H)
            t0 = time.perf_counter()
            for _ in range(5):
                out[mode] = ctx.lml(H)
            best[mode] = min(best[mode], (time.perf_counter() - t0) / 5 * 1e3)
    same = np.array_equal(out["0"], out["1"])
    print("n=%5d d=%2d B=%3d  gen %s: Gram kernel %.3f ms, generated %.3f ms (%+.1f %%), identical %s"
          % (n, d, B, ctx.gen_stats()["batches"] > 0, best["0"], best["1"], (best["1"] / best["0"] - 1) * 100, same))
    ctx.close()
