#!/usr/bin/env python3
"""cProfile of one config-E tell (Optimizer.tell, PVRS over 10 000 candidates, n ~ 977, gp_samples=128, burnin 10)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bayes_skopt_amd as bask

d, m = 8, 10000
rng = np.random.RandomState(0)
opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * d, n_points=m, n_initial_points=10, init_strategy="r2", acq_func="pvrs",
                     random_state=0)
X0 = rng.uniform(size=(974, d)).tolist()
f = lambda x: float(np.sin(3 * np.sum(x)) + 0.1 * rng.randn())
opt.tell(X0, [f(x) for x in X0], fit=False)
for it in range(3):
    x = rng.uniform(size=d).tolist()
    t0 = time.perf_counter()
    opt.tell(x, f(x), gp_samples=128, gp_burnin=10)
    print("tell %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
pr = cProfile.Profile()
x = rng.uniform(size=d).tolist()
pr.enable()
opt.tell(x, f(x), gp_samples=128, gp_burnin=10)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
