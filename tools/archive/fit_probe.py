#!/usr/bin/env python3
"""Where BayesGPR.fit() + sample() spends its wall clock at BASELINE config C (n=2048, d=16, 256 walkers x 30 steps):
the MAP start (L-BFGS-B on the device LML + gradient: sequential single-matrix evaluations) against the MCMC."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bayes_skopt_amd as bask

n, d = int(os.environ.get("FP_N", 2048)), int(os.environ.get("FP_D", 16))
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d))
y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
y = (y - y.mean()) / y.std()
for rep in range(2):
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    gp.fit(X, y, n_desired_samples=256, n_burnin=29, n_walkers_per_thread=256, progress=False)
    pr.disable()
    print("fit+sample %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
