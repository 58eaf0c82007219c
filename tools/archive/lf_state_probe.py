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
from bayes_skopt_amd import distributed
distributed.init_process_group(device=0)
print("group:", distributed.group_info(0))
show("after init_process_group (one rank)")
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
gp2 = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(16))), random_state=0, device=0)
gp2.fit(X, y, n_desired_samples=256 * 5, n_burnin=2, n_walkers_per_thread=256, progress=False)
show("after a BayesGPR.fit at n = 2048 (alive)")
del gp2
show("... deleted")
bench.config_d_roofline(_lib, 0, 76.0)
show("after config D's timed pass")
sh = bench.small_batch_shards(_lib, X, y, H[:128], 0)
show("after small_batch_shards")
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
