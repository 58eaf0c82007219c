#!/usr/bin/env python3
"""Does the launch-free path's speed depend on WHERE its buffers land?  The same call after allocations of different sizes
(kept alive, or freed again) in front of it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from bayes_skopt_amd import _lib

def show(tag):
    r = bench.launch_free(_lib, 0, shapes=((1024, 8, 32), (2048, 16, 16)))
    print(f"{tag:46s}", "  ".join(f"{k}: {v['launches_ms']:.3f} / {v['launch_free_ms']:.3f}" for k, v in r.items()), flush=True)

show("fresh")
for n, mb in ((2048, 128), (1024, 8), (4096, 8), (2048, 17), (3000, 5), (2048, 128)):
    X, y = bench.synth(n, 8, seed=1)
    c = _lib.Context(X, y, 1e-10, max_batch=mb, device=0)
    H = np.concatenate([[0.0], np.full(8, np.log(0.3)), [np.log(0.01)]]) + 0.05 * np.random.RandomState(5).randn(min(mb, 4), 10)
    c.lml(H)
    show(f"context n={n} max_batch={mb} alive")
    c.close()
    show("  ... closed")
