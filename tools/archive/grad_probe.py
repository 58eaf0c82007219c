import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
for n, d in ((128, 2), (512, 8), (1024, 8), (2048, 16)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
    ctx = _lib.Context(X, y, 1e-10, max_batch=4)
    h = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])[None, :]
    for _ in range(3): ctx.lml_grad(h)
    t0 = time.perf_counter()
    for _ in range(10): ctx.lml_grad(h)
    tg = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10): ctx.lml(h)
    tl = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10): ctx.posterior(h, want_L=False, want_alpha=True, want_K_inv=False)
    tp = (time.perf_counter() - t0) / 10
    print(f"n={n}: lml {tl*1e3:.2f} ms   posterior build {tp*1e3:.2f} ms   lml+grad {tg*1e3:.2f} ms", flush=True)
    ctx.close()
