import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayes_skopt_amd as bask
d, m = 8, 10000
for mode in ("random_x", "ask_x"):
    rng = np.random.RandomState(0)
    opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * d, n_points=m, n_initial_points=10, init_strategy="r2", acq_func="pvrs", random_state=0)
    X0 = rng.uniform(size=(974, d)).tolist()
    f = lambda x: float(np.sin(3 * np.sum(x)) + 0.1 * rng.randn())
    opt.tell(X0, [f(x) for x in X0], fit=False)
    us, pr = opt._update_surrogate, opt._propose
    T = {}
    def wrap(name, fn):
        def w(*a, **k):
            t0 = time.perf_counter(); r = fn(*a, **k); T[name] = (time.perf_counter() - t0) * 1e3; return r
        return w
    opt._update_surrogate = wrap("surrogate", us); opt._propose = wrap("propose", pr)
    out = []
    for it in range(10):
        x = opt.ask() if (mode == "ask_x" and it) else rng.uniform(size=d).tolist()
        t0 = time.perf_counter()
        opt.tell(x, f(x), gp_samples=128, gp_burnin=10)
        out.append("%.1f(s%.1f p%.1f)" % ((time.perf_counter() - t0) * 1e3, T["surrogate"], T["propose"]))
    print(mode, " ".join(out), flush=True)
