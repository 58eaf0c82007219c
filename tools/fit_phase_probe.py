#!/usr/bin/env python3
"""Where BayesGPR.fit() spends its wall clock at a BASELINE shape: fit_phase_probe.py [n d W steps]  (default: config C, 30 steps).
Phases by wrapping the methods fit() goes through: context creation, MAP start (L-BFGS-B calls of the device LML + gradient), the
sampler's start ensemble, the resident run, the posterior build."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayes_skopt_amd as bask  # noqa: E402
from bayes_skopt_amd import bayesgpr, sampler  # noqa: E402

n, d, W, steps = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (2048, 16, 256, 30)
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d))
y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
spent, calls = {}, {}


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            spent[label] = spent.get(label, 0.0) + time.perf_counter() - t0
            calls[label] = calls.get(label, 0) + 1

    setattr(obj, name, g)


G = bayesgpr.BayesGPR
wrap(G, "_ensure_context", "context (allocation + training set upload)")
wrap(G, "_map_fit", "MAP start, all of it (incl. context)")
wrap(G, "log_marginal_likelihood", "  LML + gradient calls of L-BFGS-B")
wrap(G, "sample", "sample(), all of it")
wrap(sampler.EnsembleSampler, "compute_log_prob", "  start ensemble's log-probabilities")
wrap(sampler.EnsembleSampler, "_run_resident", "  resident run")
for name in ("_build_posteriors", "_set_posterior", "_posterior_from_theta"):
    if hasattr(G, name):
        wrap(G, name, "  posterior build (%s)" % name)
for rep in range(3):
    spent.clear()
    calls.clear()
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0)
    t0 = time.perf_counter()
    gp.fit(X, y, n_desired_samples=W * (steps - 5), n_burnin=5, n_walkers_per_thread=W, progress=False)
    total = time.perf_counter() - t0
    print("fit() pass %d: %.1f ms, %d log-probability evaluations" % (rep, total * 1e3, gp._sampler.n_log_prob_evals))
    for k, v in spent.items():
        print("   %-58s %8.1f ms  (%d calls)" % (k, v * 1e3, calls[k]))
    del gp
