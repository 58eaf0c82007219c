#!/usr/bin/env python3
"""Bitwise reproducibility probe: the same inputs must give the same bits run after run and
independently of how a batch is split (no floating-point atomics on the result path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayes_skopt_amd as bask  # noqa
from bayes_skopt_amd import _lib
ok = True
for n, d in ((96, 2), (300, 3), (1024, 8), (2048, 16)):
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=32)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.1 * rng.randn(20, d + 2)
    a = ctx.lml(H)
    reruns = all(np.array_equal(a, ctx.lml(H)) for _ in range(5))
    c = np.concatenate([ctx.lml(H[:10]), ctx.lml(H[10:])])
    e = np.concatenate([ctx.lml(H[i:i+1]) for i in range(20)])
    g = [ctx.lml_grad(H[:4]) for _ in range(4)]
    grad = all(np.array_equal(g[0][1], x[1]) and np.array_equal(g[0][0], x[0]) for x in g[1:])
    ctx.posterior(H[:1])
    Xq = rng.uniform(size=(1000, d))
    pv = [ctx.predict(H[:1], Xq) for _ in range(4)]
    pred = all(np.array_equal(pv[0][0], x[0]) and np.array_equal(pv[0][1], x[1]) for x in pv[1:])
    print(n, "lml reruns", reruns, "split", np.array_equal(a, c), "single", np.array_equal(a, e), "grad", grad, "predict", pred)
    ok &= reruns and np.array_equal(a, c) and np.array_equal(a, e) and grad and pred
    ctx.close()
print("DETERMINISTIC" if ok else "NONDETERMINISTIC")
