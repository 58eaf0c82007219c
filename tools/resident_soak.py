#!/usr/bin/env python3
"""Soak of the device-resident sampler: for a set of shapes, again and again, a run with the state on the device against the same
run driven from the host (fresh seeds every round) -- positions must agree bit for bit, log-probabilities to 1e-12, no launch-free
time-out may occur.  usage: resident_soak.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayes_skopt_amd as bask  # noqa: E402
from bayes_skopt_amd.bayesgpr import _AsyncLogProb  # noqa: E402
from sklearn.gaussian_process.kernels import WhiteKernel  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
SHAPES = [(128, 2, 100, 25), (1024, 8, 64, 8), (975, 8, 100, 6), (300, 3, 40, 10), (2048, 16, 40, 4), (640, 4, 48, 8), (1536, 8, 32, 5)]
gps = []
for n, d, W, steps in SHAPES:
    rng = np.random.RandomState(n)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0, max_batch=W // 2)
    gp.kernel_ = gp.kernel + WhiteKernel(noise_level=0.01)
    gp.noise_ = 0.01
    gp.X_train_, gp.y_train_ = X, y
    gp.y_train_mean_, gp.y_train_std_ = np.zeros(1), 1
    gp._ensure_context(batch_hint=W // 2)
    gps.append((gp, bask.guess_priors(gp.kernel_)))
t_end = time.time() + budget
rounds = runs = halfsteps = 0
worst = 0.0
while time.time() < t_end:
    for (n, d, W, steps), (gp, priors) in zip(SHAPES, gps):
        seed = 1000 * rounds + n
        theta0 = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
        pos = theta0 + 1e-2 * np.random.RandomState(seed).randn(W, d + 2)
        out = []
        for resident in (False, True):
            gp.resident_sampler = resident
            smp = bask.sampler.EnsembleSampler(W, d + 2, _AsyncLogProb(gp), kwargs=dict(priors=priors))
            smp.random_state = np.random.RandomState(seed + 1).get_state()
            smp.run_mcmc(pos, steps)
            assert getattr(smp, "resident_runs", 0) == (1 if resident else 0)
            out.append((smp.get_chain(), smp.get_log_prob(), smp.naccepted.copy()))
        assert np.array_equal(out[0][0], out[1][0]), ("positions differ", n, W, rounds)
        assert np.array_equal(out[0][2], out[1][2]), ("accept counts differ", n, W, rounds)
        fin = np.isfinite(out[0][1])
        assert np.array_equal(fin, np.isfinite(out[1][1]))
        rel = np.max(np.abs(out[0][1][fin] - out[1][1][fin]) / np.abs(out[0][1][fin]))
        assert rel < 1e-12, (rel, n, W, rounds)
        worst = max(worst, float(rel))
        runs += 1
        halfsteps += 2 * steps
    rounds += 1
stats = [gp._ctx.persist_stats() for gp, _ in gps]
timeouts = sum(s["timeouts"] for s in stats)
print("resident soak: %d rounds, %d resident runs (%d half-steps) against their host-driven twins: positions and accept counts identical, "
      "log-probabilities within %.2e relative; launch-free calls %d, time-outs %d" % (
          rounds, runs, halfsteps, worst, sum(s["calls"] for s in stats), timeouts))
assert timeouts == 0
