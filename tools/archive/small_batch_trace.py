#!/usr/bin/env python3
"""lml(B=50) at n=1024, d=8 a few times: the workload traced by `rocprofv3 --kernel-trace` for tools/rocprof_gaps.py."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd  # noqa
from bayes_skopt_amd import _lib
n, d, B = int(os.environ.get("SB_N", 1024)), 8, int(os.environ.get("SB_B", 50))
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.01 * rng.randn(B, d + 2)
for _ in range(20):
    ctx.lml(H)
