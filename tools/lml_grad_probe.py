#!/usr/bin/env python3
"""N LML + gradient evaluations of ONE matrix at config C's size (what L-BFGS-B calls in the MAP start of fit()): wall clock per call;
run under rocprofv3 --kernel-trace --stats for the kernels behind it.  lml_grad_probe.py [n d calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from bayes_skopt_amd import _lib  # noqa: E402

n, d, calls = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 16, 40)
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d))
y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
ctx = _lib.Context(X, y, 1e-10, max_batch=128)
h = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])[None, :]
ctx.lml_grad(h)
t0 = time.perf_counter()
for i in range(calls):
    ctx.lml_grad(h + 1e-3 * i)
print("%.3f ms per LML + gradient call (n = %d, d = %d)" % ((time.perf_counter() - t0) / calls * 1e3, n, d))
ctx.close()
