#!/usr/bin/env python3
"""Run a few LML batch calls at one shape (for rocprofv3 timelines): one_shape_probe.py n d B [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd  # noqa
from bayes_skopt_amd import _lib
n, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.01 * rng.randn(B, d + 2)
for _ in range(reps): ctx.lml(H)
ctx.close()
