#!/usr/bin/env python3
"""BASELINE config D alone (n = 4096, d = 32, blocked fp64 Cholesky with the MFMA trailing update, 8 matrices per batch):
what tools/profile_config_d.sh runs under rocprofv3, so that the kernel table and the PMC passes hold this
configuration only.  Prints one JSON line: wall per batch and the trailing update's algorithmic TFLOP/s from HIP events
(the same arithmetic as bench.py's `roofline_n4096`).  BGP_STREAMS=1 for per-kernel numbers."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd  # noqa: E402,F401
from bayes_skopt_amd import _lib  # noqa: E402


def main():
    n, d, B = 4096, 32, 8
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    ctx = _lib.Context(X, y, 1e-10, max_batch=B)
    ctx.set_streams(1)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * np.random.RandomState(3).randn(B, d + 2)
    ctx.lml(H)
    t0 = time.perf_counter()
    for _ in range(reps):
        v = ctx.lml(H)
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.set_timing(True)
    syrk_ms = launches = 0
    for _ in range(3):
        ctx.lml(H)
        tm = ctx.last_timing()
        syrk_ms += tm["syrk"]["ms"]
        launches += tm["syrk"]["launches"]
    flops = float(sum(128 * (n - j * 128) * (n - j * 128 + 1) for j in range(1, n // 128))) * B * 3
    print(json.dumps({"config": "D", "n": n, "d": d, "batch": B, "ms_per_batch": wall,
                      "syrk_tflops_hip_events": flops / (syrk_ms * 1e-3) / 1e12, "syrk_launches_per_batch": launches // 3,
                      "syrk_avg_launch_ms": syrk_ms / launches, "algorithmic_flops_per_matrix": flops / (3 * B),
                      "lml0": float(v[0])}))
    ctx.close()


if __name__ == "__main__":
    main()
