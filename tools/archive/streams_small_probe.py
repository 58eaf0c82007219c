#!/usr/bin/env python3
"""Walker-group streams at SMALL batches: wall time per LML call for 1 / 2 / 4 / 8 groups (each group = a slice of the
batch on its own HIP stream; one group's latency-bound potrf chain can run under another group's MFMA-bound update),
with a bit-identity check against the one-group result.  GPU box only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd  # noqa: E402,F401
from bayes_skopt_amd import _lib  # noqa: E402


def run(n, d, B, groups=(1, 2, 4, 8), reps=40):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.01 * rng.randn(B, d + 2)
    ref = None
    out = []
    for g in groups:
        if B // g < 8:
            continue
        ctx.set_streams(g)
        for _ in range(5):
            v = ctx.lml(H)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            v = ctx.lml(H)
            ts.append(time.perf_counter() - t0)
        if ref is None:
            ref = v.copy()
        out.append(f"g={g}: {np.median(ts) * 1e3:.3f} ms (min {np.min(ts) * 1e3:.3f}) same={bool(np.array_equal(v, ref))}")
    print(f"n={n} d={d} B={B}:  " + "   ".join(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    for n, d, B in ((1024, 8, 32), (1024, 8, 64), (975, 8, 50), (2048, 16, 16), (2048, 16, 32), (2048, 16, 64),
                    (2048, 16, 128), (4096, 32, 8), (4096, 32, 16)):
        run(n, d, B)
