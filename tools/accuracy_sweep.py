#!/usr/bin/env python3
"""Worst-case relative error of the device LML / predict against the CPU oracle over a grid of
length scales and noise levels (conditioning from benign to ~1e12).  GPU box only (test/measurement tool)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd  # noqa
from bayes_skopt_amd import _lib
from oracle import gp_oracle as O

def synth(n, d, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    return X, (y - y.mean()) / y.std()

for n, d in ((512, 2), (1024, 8)):
    X, y = synth(n, d, 0)
    ad = np.full(n, 1e-10)
    ctx = _lib.Context(X, y, ad, max_batch=16)
    Xq = np.random.RandomState(1).uniform(size=(64, d))
    print(f"n={n} d={d}:  ell   s2      cond(K)    rel.err LML   rel.err mean   abs.err std")
    for ell in (0.1, 0.3, 1.0, 3.0):
        for s2 in (1e-8, 1e-6, 1e-4, 1e-2):
            h = np.concatenate([[0.0], np.full(d, np.log(ell)), [np.log(s2)]])
            K = O.gram_with_jitter(X, ad, h)
            cond = np.linalg.cond(K)
            ref = O.lml(X, y, ad, h)
            got = ctx.lml(h)[0]
            ctx.posterior(h)
            mean, var = ctx.predict(h, Xq)
            mo, so = O.predict(X, y, ad, h, Xq)
            print(f"          {ell:4.1f}  {s2:7.0e}  {cond:9.2e}   {abs(got-ref)/abs(ref):9.2e}    "
                  f"{np.max(np.abs(mean[0]-mo))/max(np.max(np.abs(mo)),1e-300):9.2e}     {np.max(np.abs(np.sqrt(var[0])-so)):9.2e}")
    ctx.close()
