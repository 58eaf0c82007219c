#!/usr/bin/env python3
"""Exploratory GPU probe: micro-benchmarks + per-kernel timing of the LML batch at the BASELINE configs."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd as bask  # noqa: E402
from bayes_skopt_amd import _lib  # noqa: E402


def synth(n, d, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    return X, (y - y.mean()) / y.std()


def thetas(d, B, seed, spread=0.2):
    rng = np.random.RandomState(seed)
    base = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
    return base + spread * rng.randn(B, d + 2)


def main():
    out = {}
    print("devices:", _lib.device_count())
    rows, cols = _lib.mfma_f64_layout()
    lane = np.arange(64)[:, None]
    reg = np.arange(4)[None, :]
    print("layout row==(lane>>4)+4*reg:", bool(np.all(rows == (lane >> 4) + 4 * reg)),
          " col==lane&15:", bool(np.all(cols == (lane & 15))))
    if not np.all(rows == (lane >> 4) + 4 * reg):
        print("rows:\n", rows[:20])
    out["mfma_f64_tflops"] = _lib.bench_mfma_f64(iters=20000)
    out["hbm_copy_gbps"] = _lib.bench_hbm_copy(nbytes=1 << 30, iters=10)
    print(out)
    cases = [(128, 2, 50), (1024, 8, 32), (2048, 16, 128), (4096, 32, 8)]
    if len(sys.argv) > 1:
        cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for n, d, B in cases:
        X, y = synth(n, d, 0)
        ctx = _lib.Context(X, y, 1e-10, max_batch=B)
        H = thetas(d, B, 30)
        ctx.lml(H)  # warm-up
        for ns in (1, 2, 3, 4, 8):
            ctx.set_streams(ns)
            ctx.lml(H)
            t0 = time.perf_counter()
            reps = 4
            for _ in range(reps):
                v = ctx.lml(H)
            dt = (time.perf_counter() - t0) / reps
            print(f"   streams={ns}: wall {dt*1e3:.3f} ms/batch ({B/dt:.1f} evals/s)")
        ctx.set_streams(1)
        ctx.set_timing(True)
        ctx.lml(H)
        tm = ctx.last_timing()
        ctx.set_timing(False)
        nb = (n + 127) // 128
        f_trail = sum(128 * (n_ - 0) * 0 for n_ in [0])  # placeholder
        m = [(nb - j) * 128 for j in range(1, nb)]
        f_trail = sum(128 * mj * (mj + 128) for mj in m) * B  # 2 flop per MAC, lower tiles incl. full diagonal tiles
        f_chol = (n**3 / 3 + n**2 / 2 + n / 6) * B
        syrk_ms = tm["syrk"]["ms"]
        print(f"n={n} d={d} B={B}: wall {dt*1e3:.2f} ms/batch  ({B/dt:.1f} evals/s)  lml[0]={v[0]:.6f}")
        print("   timing:", json.dumps(tm))
        if syrk_ms > 0:
            print(f"   syrk: {f_trail/syrk_ms/1e9:.2f} TFLOP/s (algorithmic tile flops);  "
                  f"chol total {f_chol/(tm['potrf']['ms']+tm['trsm']['ms']+syrk_ms)/1e9:.2f} TFLOP/s")
        ctx.close()


if __name__ == "__main__":
    main()
