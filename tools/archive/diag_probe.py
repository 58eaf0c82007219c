#!/usr/bin/env python3
"""Monte-Carlo spread of the post-hoc diagnostics on the reference's five-point problem (tests/test_optimizer.py:85-141
of the reference pins 0.99 / (0.98, 0.86) / 0.99 and 0.3 / 0.25 / 0.29 from ONE realisation each)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bayes_skopt_amd as bask


def five(seed):
    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=0, random_state=np.random.RandomState(seed))
    opt.tell([[-2.0], [-1.0], [0.0], [1.0], [2.0]], [2.0, 0.0, -2.0, 0.0, 2.0], gp_burnin=10)
    return opt


opt = five(0)
for kw in (dict(normalized_scores=False, use_mean_gp=True), dict(normalized_scores=True, use_mean_gp=True),
           dict(normalized_scores=True, use_mean_gp=False)):
    gaps = [opt.expected_optimality_gap(random_state=np.random.RandomState(seed), n_probabilities=10, n_space_samples=100,
                                        n_gp_samples=100, n_random_starts=10, tol=0.1, **kw) for seed in range(12)]
    print(kw, "mean6 %.3f mean12 %.3f sd %.3f" % (np.mean(gaps[:6]), np.mean(gaps), np.std(gaps)), np.round(gaps, 3), flush=True)
