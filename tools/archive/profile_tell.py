#!/usr/bin/env python3
"""cProfile of one Optimizer.tell at config-E size (n~980, 10k candidates, PVRS)."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd as bask
rng = np.random.RandomState(0)
d, n0, m = 8, 974, 10000
f = lambda x: float(np.sin(3 * np.sum(x)) + 0.1 * rng.randn())
opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * d, n_points=m, n_initial_points=10, init_strategy="r2", acq_func="pvrs", random_state=0)
X0 = rng.uniform(size=(n0, d)).tolist()
opt.tell(X0, [f(x) for x in X0], fit=False)
x = rng.uniform(size=d).tolist(); opt.tell(x, f(x), gp_samples=128, gp_burnin=10)
x = opt.ask(); t0 = time.perf_counter(); opt.tell(x, f(x), gp_samples=128, gp_burnin=10); print("warm tell ms", 1e3 * (time.perf_counter() - t0))
pr = cProfile.Profile(); x = opt.ask(); pr.enable(); opt.tell(x, f(x), gp_samples=128, gp_burnin=10); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
