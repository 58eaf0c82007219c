import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import bayes_skopt_amd as bask

def five(seed):
    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=0, random_state=np.random.RandomState(seed))
    opt.tell([[-2.0], [-1.0], [0.0], [1.0], [2.0]], [2.0, 0.0, -2.0, 0.0, 2.0], gp_burnin=10)
    return opt
for ngp in (200, 2000):
    for seed in (0, 1, 2):
        opt = five(0)
        t0 = time.time()
        p1 = opt.probability_of_optimality(threshold=1.0, n_random_starts=100, random_state=np.random.RandomState(seed), normalized_scores=False, n_gp_samples=ngp)
        p2 = opt.probability_of_optimality(threshold=(0.9, 0.5), n_random_starts=100, random_state=np.random.RandomState(seed), normalized_scores=False, n_gp_samples=ngp)
        p3 = opt.probability_of_optimality(threshold=1.0, n_random_starts=100, random_state=np.random.RandomState(seed), normalized_scores=True, n_gp_samples=ngp)
        print("ngp", ngp, "seed", seed, "prob", p1, p2, p3, "%.2fs" % (time.time() - t0), flush=True)
for ngp, nsp, npb in ((100, 100, 10), (2000, 500, 50)):
    for seed in (0, 1, 2):
        opt = five(0)
        t0 = time.time()
        gaps = []
        for kw in (dict(normalized_scores=False, use_mean_gp=True), dict(normalized_scores=True, use_mean_gp=True), dict(normalized_scores=True, use_mean_gp=False)):
            gaps.append(opt.expected_optimality_gap(random_state=np.random.RandomState(seed), n_probabilities=npb, n_space_samples=nsp, n_gp_samples=ngp, n_random_starts=10, tol=0.1 if ngp == 100 else 0.01, **kw))
        print("gap ngp", ngp, "seed", seed, gaps, "%.2fs" % (time.time() - t0), flush=True)
