#!/usr/bin/env python3
"""Soak of the device-resident sampler: for a set of shapes, again and again, a run with the state on the device against the same
run driven from the host (fresh seeds every round) -- positions must agree bit for bit, log-probabilities to 1e-12, no launch-free
time-out may occur.  Round 6: the shapes include walkers that carry their own input warp (warp_inputs=True), an ensemble that takes
the HBM form of the step kernel, and -- `resident_soak.py <seconds> loopback` -- the sharded run over a loop-back group of 2 / 3
ranks (threads of this process on the one GPU) against the single-context run.  usage: resident_soak.py [seconds] [loopback]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayes_skopt_amd as bask  # noqa: E402
from bayes_skopt_amd.bayesgpr import _AsyncLogProb  # noqa: E402
from sklearn.gaussian_process.kernels import WhiteKernel  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
if len(sys.argv) > 2 and sys.argv[2] == "loopback":
    # the sharded resident run (tests/test_gpu_resident.py::_loopback_run) over and over with fresh seeds
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_gpu_resident import _loopback_run

    cases = [(2, 300, 3, 40, 10, 0, False), (3, 260, 2, 50, 6, 0, False), (2, 200, 2, 24, 6, 0, True), (2, 1024, 4, 32, 4, 0, False)]
    t_end, rounds, runs = time.time() + budget, 0, 0
    while time.time() < t_end:
        for world, n, d, W, steps, persist, warp in cases:
            ref, results = _loopback_run(world, n, d, W, steps, persist, warp, seed=100 * rounds + n)
            for got in results:
                for a, b in zip(ref[:5], got[:5]):
                    assert np.array_equal(a, b), (world, n, rounds)
            runs += world
        rounds += 1
    print("loop-back soak: %d rounds, %d rank runs of the sharded resident sampler identical to the single-context run bit for bit "
          "(chain, log-probabilities, final ensemble, accept counts)" % (rounds, runs))
    sys.exit(0)

import scipy.stats as st  # noqa: E402

# (n, d, walkers, steps, warped walkers)
SHAPES = [(128, 2, 100, 25, False), (1024, 8, 64, 8, False), (975, 8, 100, 6, False), (300, 3, 40, 10, False), (2048, 16, 40, 4, False),
          (640, 4, 48, 8, False), (1536, 8, 32, 5, False), (300, 3, 40, 8, True), (100, 2, 24, 12, True), (200, 40, 256, 2, False),
          # the launch schedule with the Gram blocks generated inside the trailing update (resident run) against the Gram kernel in
          # front (host-driven twin, BGP_SYRK_GEN=0): BASELINE config C's ensemble, and a two-panel-group shape
          (2048, 16, 256, 2, False), (1100, 8, 256, 3, False)]
gps = []
for n, d, W, steps, warp in SHAPES:
    rng = np.random.RandomState(n)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0, max_batch=W // 2, warp_inputs=warp)
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
    for (n, d, W, steps, warp), (gp, priors) in zip(SHAPES, gps):
        seed = 1000 * rounds + n
        theta0 = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)], np.zeros(2 * d if warp else 0)])
        p = len(theta0)
        pos = theta0 + 1e-2 * np.random.RandomState(seed).randn(W, p)
        wp = (st.norm(loc=0.0, scale=0.3).logpdf, st.norm(loc=0.0, scale=0.3).logpdf)
        out = []
        for resident in (False, True):
            os.environ["BGP_SYRK_GEN"] = "1" if resident else "0"
            gp.resident_sampler = resident
            smp = bask.sampler.EnsembleSampler(W, p, _AsyncLogProb(gp), kwargs=dict(priors=priors, warp_priors=wp))
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
os.environ.pop("BGP_SYRK_GEN", None)
generated = sum(gp._ctx.gen_stats()["batches"] for gp, _ in gps)
stats = [gp._ctx.persist_stats() for gp, _ in gps]
timeouts = sum(s["timeouts"] for s in stats)
print("resident soak: %d rounds, %d resident runs (%d half-steps) against their host-driven twins: positions and accept counts identical, "
      "log-probabilities within %.2e relative; launch-free calls %d, time-outs %d; LML batches with their Gram blocks generated inside the "
      "trailing update %d" % (rounds, runs, halfsteps, worst, sum(s["calls"] for s in stats), timeouts, generated))
assert timeouts == 0
