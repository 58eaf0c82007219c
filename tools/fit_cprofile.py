#!/usr/bin/env python3
"""cProfile of one warm BayesGPR.fit() at config C (host-side functions by cumulative time): fit_cprofile.py [n d W steps]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayes_skopt_amd as bask  # noqa: E402

n, d, W, steps = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (2048, 16, 256, 30)
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d))
y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
for rep in range(2):
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0)
    pr = cProfile.Profile()
    pr.enable()
    gp.fit(X, y, n_desired_samples=W * (steps - 5), n_burnin=5, n_walkers_per_thread=W, progress=False)
    pr.disable()
    if rep == 1:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
