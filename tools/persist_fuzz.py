#!/usr/bin/env python3
"""Random shapes through both factorisation paths: n, d, batch size, kernel family and form, a failing matrix now and then --
log-likelihoods, statuses, the factor and z of one slot must be the same bits on the launch schedule and launch-free.
usage: persist_fuzz.py [cases=60] [seed=0]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayes_skopt_amd  # noqa: E402,F401
from bayes_skopt_amd import _lib  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    n = int(rng.choice([rng.randint(129, 700), rng.randint(700, 1400), rng.randint(1400, 2600)]))
    d = int(rng.randint(1, 21))
    B = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 31, 32, 33, 48, 50, 64]))
    if n > 1400 and B > 32:
        B = int(rng.randint(1, 33))
    form = rng.choice(["product", "sum"])
    stat = rng.choice(["rbf", "matern12", "matern32", "matern52"])
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    ctx = _lib.Context(X, y, np.full(n, 1e-10), form=form, stationary=stat, max_batch=B)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.1 * rng.randn(B, d + 2)
    if B > 2 and rng.rand() < 0.3:
        H[rng.randint(B), 0] = np.nan  # fails at the first pivot
    if B > 2 and rng.rand() < 0.3:
        k = rng.randint(B)
        H[k, 1:d + 1] = 9.0   # flat kernel, no noise: fails somewhere inside
        H[k, d + 1] = -40.0
    out = {}
    for tag, mode in (("launches", 0), ("launch_free", 1)):
        ctx.set_persist(mode)
        v, st = ctx.lml(H, return_status=True)
        L, z = ctx.debug_workspace(int(rng.randint(B)) if False else 0)
        out[tag] = (v.copy(), st.copy(), np.tril(L).copy(), z.copy())
    a, b = out["launches"], out["launch_free"]
    ok_slot0 = a[1][0] != 0 or (np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]))
    same = np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1]) and ok_slot0
    bad += 0 if same else 1
    print(f"{it:3d} n={n:5d} d={d:2d} B={B:2d} {stat:9s} {form:7s} failed {int((a[1] != 0).sum()):2d}  {'same bits' if same else 'DIFFERENT'}", flush=True)
    ctx.close()
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
