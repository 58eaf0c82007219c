#!/usr/bin/env python3
"""The reference's pinned post-hoc diagnostics (tests/test_optimizer.py:85-141 of kiudee/bayes-skopt: RandomState(123) shared by
the Optimizer and the diagnostic call) with reference-variate function draws (BayesGPR.mvn = "reference") and with the device
Cholesky draws: prints both next to the reference's two-decimal pins."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayes_skopt_amd as bask

def make(mode):
    rs = np.random.RandomState(123)
    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=0, random_state=rs)
    opt.tell([[-2.0], [-1.0], [0.0], [1.0], [2.0]], [2.0, 0.0, -2.0, 0.0, 2.0], gp_burnin=10)
    opt.gp.mvn = mode
    return opt, rs

for mode in ("reference", "cholesky"):
    for kw, pin in ((dict(normalized_scores=False, threshold=1.0), 0.99),
                    (dict(normalized_scores=False, threshold=(0.9, 0.5)), (0.98, 0.86)),
                    (dict(normalized_scores=True, threshold=1.0), 0.99)):
        opt, rs = make(mode)
        t0 = time.perf_counter()
        p = opt.probability_of_optimality(threshold=kw["threshold"], n_random_starts=100, random_state=rs,
                                          normalized_scores=kw["normalized_scores"])
        print(mode, "prob", kw, "->", p, "pin", pin, "%.2f s" % (time.perf_counter() - t0), flush=True)
    for kw, pin in ((dict(normalized_scores=False, use_mean_gp=True), 0.3), (dict(normalized_scores=True, use_mean_gp=True), 0.25),
                    (dict(normalized_scores=True, use_mean_gp=False), 0.29)):
        opt, rs = make(mode)
        t0 = time.perf_counter()
        g = opt.expected_optimality_gap(random_state=rs, n_probabilities=10, n_space_samples=100, n_gp_samples=100,
                                        n_random_starts=10, tol=0.1, **kw)
        print(mode, "gap", kw, "->", round(g, 4), "pin", pin, "%.2f s" % (time.perf_counter() - t0), flush=True)
