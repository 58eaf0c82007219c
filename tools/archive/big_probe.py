import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
for n, d in ((12288, 4), (20000, 3)):
    rng = np.random.RandomState(1)
    X = rng.uniform(size=(n, d)); y = rng.randn(n)
    ctx = _lib.Context(X, y, np.full(n, 1e-3), max_batch=1)
    # (a) white-dominated closed form
    c, s2 = np.exp(-42.0), 4.0
    h = np.concatenate([[-42.0], np.full(d, np.log(0.3)), [np.log(s2)]])
    t0 = time.perf_counter(); got = ctx.lml(h)[0]; dt = time.perf_counter() - t0
    d0 = c + s2 + 1e-3
    ref = -0.5 * float(y @ y) / d0 - 0.5 * n * np.log(d0) - 0.5 * n * np.log(2 * np.pi)
    # (b) a real kernel: block-diagonal additivity is not available here; compare two permutations
    h2 = np.concatenate([[0.0], np.full(d, np.log(0.05)), [np.log(0.1)]])
    t1 = time.perf_counter(); v1 = ctx.lml(h2)[0]; dt2 = time.perf_counter() - t1
    ctx.close()
    perm = rng.permutation(n)
    ctx = _lib.Context(X[perm], y[perm], np.full(n, 1e-3), max_batch=1)
    v2 = ctx.lml(h2)[0]
    ctx.close()
    print(f"n={n}: white closed form rel err {abs(got-ref)/abs(ref):.2e} ({dt*1e3:.0f} ms)  permutation invariance rel diff {abs(v1-v2)/abs(v1):.2e}  ({dt2*1e3:.0f} ms per LML, {n**3/3/dt2/1e12:.1f} TF)")
