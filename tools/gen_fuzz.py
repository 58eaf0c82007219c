#!/usr/bin/env python3
"""Random shapes through the launch schedule with the Gram blocks generated inside the trailing update (the default) and with the
Gram kernel in front (BGP_SYRK_GEN=0): the log-likelihoods must be bit-identical.  gen_fuzz.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from bayes_skopt_amd import _lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, cases, generated, worst = time.time(), 0, 0, 0.0
kinds = ["rbf", "matern12", "matern32", "matern52"]
while time.time() - t0 < budget:
    nblk = int(rs.randint(2, 14))
    n = int(min(128 * nblk - rs.choice([0, 0, 1, 63, 64, 65, 127, rs.randint(0, 128)]), 128 * nblk))
    n = max(n, 128 * (nblk - 1) + 1)
    d = int(rs.randint(1, 17))
    tri = nblk * (nblk + 1) // 2
    B = int(min(max(-(-2048 // tri), 8) + rs.randint(0, 24), 160))
    st, form = kinds[rs.randint(4)], ["product", "sum"][rs.randint(2)]
    warp = rs.rand() < 0.2
    X = rs.uniform(size=(n, d))
    if rs.rand() < 0.3:
        X[rs.randint(n)] = X[rs.randint(n)]  # a duplicated point
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rs.randn(n)
    H = np.concatenate([[rs.uniform(-1, 1)], np.log(rs.uniform(0.1, 2.0, size=d)), [np.log(rs.uniform(1e-4, 0.5))]]) \
        + 0.2 * rs.randn(B, d + 2)
    if rs.rand() < 0.3:
        H[rs.randint(B), -1] = -np.inf
    W = 0.3 * rs.randn(B, 2 * d) if warp else None
    ctx = _lib.Context(X, y, 10.0 ** rs.uniform(-12, -6), form=form, stationary=st, max_batch=B)
    ctx.set_persist(0)
    ctx.set_streams(int(rs.choice([1, 2])))
    out = {}
    for mode in ("0", "1"):
        os.environ["BGP_SYRK_GEN"] = mode
        out[mode] = ctx.lml_warped(H, W) if warp else ctx.lml(H)
    g = ctx.gen_stats()["batches"]
    ctx.close()
    cases += 1
    generated += g > 0
    same = np.array_equal(out["0"], out["1"])
    if not same:
        print("MISMATCH n=%d d=%d B=%d %s %s warp=%s: max |diff| %.3e" % (n, d, B, st, form, warp, np.nanmax(np.abs(out["0"] - out["1"]))))
        sys.exit(1)
print("gen fuzz: %d random cases in %.0f s (%d of them generated inside the update): log-likelihoods bit-identical to the Gram kernel's"
      % (cases, time.time() - t0, generated))
