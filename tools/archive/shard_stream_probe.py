#!/usr/bin/env python3
"""Wall time of one LML call for the per-GPU shares of config C's strong-scaling split (16 / 32 / 64 matrices of n = 2048, d = 16)
under whatever BGP_STREAMS / BGP_PERSIST the environment sets.  usage: BGP_STREAMS=2 BGP_PERSIST=0 shard_stream_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
n, d = 2048, 16
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
for B in (16, 32, 64):
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * rng.randn(B, d + 2)
    for _ in range(4): ctx.lml(H)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); ctx.lml(H); ts.append(time.perf_counter() - t0)
    print("B=%d %.3f ms" % (B, np.median(ts) * 1e3), end="   ")
    ctx.close()
print()
