#!/usr/bin/env python3
"""Why is the launch-free path ~8 % slower late in bench.py than in a fresh process?  launch_free() after one step at a time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import bayes_skopt_amd as bask
from bayes_skopt_amd import _lib

def show(tag):
    r = bench.launch_free(_lib, 0, shapes=((2048, 16, 16), (1024, 8, 32)))
    print(f"{tag:40s}", "  ".join(f"{k}: {v['launches_ms']:.3f} / {v['launch_free_ms']:.3f}" for k, v in r.items()), flush=True)

show("fresh")
X, y = bench.synth(2048, 16, seed=0)
ctx = _lib.Context(X, y, 1e-10, max_batch=128, device=0)
H = np.concatenate([[0.0], np.full(16, np.log(0.3)), [np.log(0.01)]]) + 0.05 * np.random.RandomState(5).randn(128, 18)
for _ in range(20):
    ctx.lml(H)
show("big context alive, 20 calls of 128")
ctx.set_streams(1); ctx.set_timing(True)
for _ in range(5):
    ctx.lml(H)
ctx.set_timing(False)
show("after a timed pass on it")
ctx.close()
show("big context closed")
print("mfma", _lib.bench_mfma_f64(0))
show("after the MFMA peak probe")
t0 = time.time()
a = np.random.rand(3000, 3000); (a @ a).sum()
show("after a host BLAS call")
keep = []
for i in range(6):
    Xs, ys = bench.synth(1024, 8, seed=i)
    c = _lib.Context(Xs, ys, 1e-10, max_batch=32, device=0)
    Hs = np.concatenate([[0.0], np.full(8, np.log(0.3)), [np.log(0.01)]]) + 0.05 * np.random.RandomState(5).randn(32, 10)
    c.set_persist(1)
    c.lml(Hs); c.lml(Hs[:8]); c.lml(Hs[:16])
    keep.append(c)
    show(f"{i + 1} more contexts alive that used the path")
for c in keep:
    c.close()
show("all of them closed")
