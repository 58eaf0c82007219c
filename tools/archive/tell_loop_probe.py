#!/usr/bin/env python3
"""Wall time per Optimizer.tell over a growing data set (the regime of a real tuning run: n = 10 .. 150,
default n_points / gp_samples / burn-in) and a cProfile of the last 20 tells."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd as bask

rng = np.random.RandomState(0)
d = 4
f = lambda x: float(np.sin(3 * np.sum(x)) + 0.1 * rng.randn())
opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * d, n_initial_points=10, random_state=0)
times = []
prof = cProfile.Profile()
for it in range(150):
    x = opt.ask()
    y = f(x)
    if it == 130:
        prof.enable()
    t0 = time.perf_counter()
    opt.tell(x, y)
    times.append(time.perf_counter() - t0)
prof.disable()
t = np.array(times) * 1e3
for lo, hi in ((0, 10), (10, 20), (20, 50), (50, 100), (100, 150)):
    print(f"tells {lo:3d}..{hi:3d}: median {np.median(t[lo:hi]):7.2f} ms  max {np.max(t[lo:hi]):7.2f} ms")
pstats.Stats(prof).sort_stats("cumulative").print_stats(18)
