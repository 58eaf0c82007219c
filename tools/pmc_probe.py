#!/usr/bin/env python3
"""One LML batch at config C under rocprofv3 --pmc (few launches, quick)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd  # noqa
from bayes_skopt_amd import _lib
n, d, B = 2048, 16, 128
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d)); y = np.sin(3 * X.sum(1)); y = (y - y.mean()) / y.std()
H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.2 * np.random.RandomState(30).randn(B, d + 2)
ctx = _lib.Context(X, y, 1e-10, max_batch=B)
ctx.lml(H); ctx.lml(H)
