#!/usr/bin/env python3
"""Where the time of the config-E EI variant goes (128 hyper-posterior samples x 10 000 candidates at n ~ 1000):
posterior batch build, batched predict, and the pieces of a whole tell."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bayes_skopt_amd  # noqa: F401
from bayes_skopt_amd import _lib

n, d, m, B = 975, 8, 10000, 128
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d))
y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(n)
y = (y - y.mean()) / y.std()
Xq = rng.uniform(size=(m, d))
TH = np.concatenate([[0.0], np.full(d, np.log(0.4)), [np.log(0.02)]]) + 0.1 * rng.randn(B, d + 2)
ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=64)
for rep in range(3):
    t0 = time.perf_counter()
    res = ctx.posterior(TH, want_alpha=False)
    t1 = time.perf_counter()
    TH0 = TH.copy()
    TH0[:, -1] = -np.inf
    mean, var = ctx.predict(TH0, Xq)
    t2 = time.perf_counter()
    print("posterior(128) %.1f ms   predict(128 x 10k) %.1f ms (%.1f TF on 2 m n^2)" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, B * 2.0 * m * 1024 * 1024 / (t2 - t1) / 1e12), flush=True)
t0 = time.perf_counter()
for _ in range(5):
    ctx.lml(TH[:50])
print("lml(50) %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))

# ---- a whole tell of the EI variant under cProfile
import cProfile
import pstats

import bayes_skopt_amd as bask

rng = np.random.RandomState(0)
opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * d, n_points=m, n_initial_points=10, init_strategy="r2", acq_func="ei",
                     random_state=0)
X0 = rng.uniform(size=(974, d)).tolist()
f = lambda x: float(np.sin(3 * np.sum(x)) + 0.1 * rng.randn())
opt.tell(X0, [f(x) for x in X0], fit=False)
for it in range(2):
    x = rng.uniform(size=d).tolist()
    opt.tell(x, f(x), gp_samples=200, gp_burnin=10, n_samples=128)
pr = cProfile.Profile()
x = rng.uniform(size=d).tolist()
t0 = time.perf_counter()
pr.enable()
opt.tell(x, f(x), gp_samples=200, gp_burnin=10, n_samples=128)
pr.disable()
print("tell %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
