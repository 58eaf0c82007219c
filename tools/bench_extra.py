#!/usr/bin/env python3
"""Extra measurements for DESIGN.md / BASELINE configs B, C (fit+sample wall-clock), D, E.
Prints one JSON object per section.  GPU box only.  Like bench.py's cpu_baseline leg, the oracle is imported
here ONLY to time the CPU restatement next to the device path (never as part of what is measured as "GPU")."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_skopt_amd as bask  # noqa: E402
from bayes_skopt_amd import _lib  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402  (CPU baseline leg only)


def synth(n, d, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    return X, (y - y.mean()) / y.std()


def cpu_eval_time(X, y, thetas, budget=8.0):
    ad = np.full(len(y), 1e-10)
    O.lml(X, y, ad, thetas[0])
    t0 = time.perf_counter()
    k = 0
    for th in thetas:
        O.lml(X, y, ad, th)
        k += 1
        if time.perf_counter() - t0 > budget:
            break
    return (time.perf_counter() - t0) / k


def fit_sample(n, d, W, steps, burnin, tag):
    X, y = synth(n, d, 0)
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0)
    t0 = time.perf_counter()
    gp.fit(X, y, n_desired_samples=W * steps, n_burnin=burnin, n_walkers_per_thread=W, progress=False)
    t_fit = time.perf_counter() - t0
    evals = gp._sampler.n_log_prob_evals
    t1 = time.perf_counter()
    gp.sample(n_desired_samples=W * steps, n_burnin=burnin, n_walkers_per_thread=W)
    t_sample = time.perf_counter() - t1
    cpu_t = cpu_eval_time(X, y, gp.chain_[:32])
    out = {
        "section": tag, "n": n, "d": d, "walkers": W, "steps": steps + burnin,
        "gpu_fit_plus_sample_ms": 1e3 * t_fit, "gpu_sample_only_ms": 1e3 * t_sample, "mcmc_evals": int(evals),
        "gpu_evals_per_s_sample": evals / t_sample,
        "cpu_ms_per_lml_eval": 1e3 * cpu_t,
        "cpu_fit_plus_sample_ms_estimate": 1e3 * cpu_t * evals,
        "speedup_estimate": cpu_t * evals / t_fit,
        "median_theta": gp.theta.tolist(), "lml_at_median": gp.log_marginal_likelihood_value_,
    }
    print(json.dumps(out), flush=True)
    return gp


def config_d():
    n, d = 4096, 32
    X, y = synth(n, d, 0)
    base = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
    for B in (1, 8):
        ctx = _lib.Context(X, y, 1e-10, max_batch=B)
        H = base + 0.2 * np.random.RandomState(30).randn(B, d + 2)
        for _ in range(3):
            ctx.lml(H)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            v = ctx.lml(H)
            ts.append(time.perf_counter() - t0)
        dt = float(np.median(ts))
        ctx.set_timing(True)
        ctx.lml(H)
        tm = ctx.last_timing()
        fl = sum(128 * (n - j * 128) * (n - j * 128 + 1) for j in range(1, n // 128)) * B
        print(json.dumps({"section": "D", "n": n, "d": d, "batch": B, "ms_per_batch": 1e3 * dt,
                          "syrk_ms": tm["syrk"]["ms"], "syrk_algorithmic_tflops": fl / tm["syrk"]["ms"] / 1e9,
                          "potrf_ms": tm["potrf"]["ms"], "trsm_ms": tm["trsm"]["ms"], "kbuild_ms": tm["kbuild"]["ms"],
                          "lml0": float(v[0])}), flush=True)
        ctx.close()


def config_e(n_iter=10, n0=974, m=10000, d=8):
    rng = np.random.RandomState(0)

    def f(x):
        return float(np.sin(3 * np.sum(x)) + 0.1 * rng.randn())

    opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * d, n_points=m, n_initial_points=10, init_strategy="r2",
                         acq_func="pvrs", random_state=0)
    X0 = rng.uniform(size=(n0, d)).tolist()
    opt.tell(X0, [f(x) for x in X0], fit=False)
    times = []
    for it in range(n_iter):
        x = opt.ask() if opt._next_x is not None else rng.uniform(size=d).tolist()
        t0 = time.perf_counter()
        opt.tell(x, f(x), gp_samples=128, gp_burnin=10, n_samples=0)
        times.append(time.perf_counter() - t0)
    print(json.dumps({"section": "E-pvrs", "n_start": n0 + 1, "candidates": m, "iters": n_iter,
                      "first_tell_ms(fit)": 1e3 * times[0], "median_tell_ms(sample)": 1e3 * float(np.median(times[1:])),
                      "evals_per_tell": 100 + 12 * 100}), flush=True)
    # EI variant: 128 hyper-posterior samples x 10k-candidate predict (batched posterior build + predict)
    opt2 = bask.Optimizer(dimensions=[(0.0, 1.0)] * d, n_points=m, n_initial_points=10, init_strategy="r2",
                          acq_func="ei", random_state=0)
    opt2.tell(X0, [f(x) for x in X0], fit=False)
    times2 = []
    for it in range(3):
        x = rng.uniform(size=d).tolist()
        t0 = time.perf_counter()
        opt2.tell(x, f(x), gp_samples=200, gp_burnin=10, n_samples=128)
        times2.append(time.perf_counter() - t0)
    print(json.dumps({"section": "E-ei128", "tell_ms": [1e3 * t for t in times2]}), flush=True)
    # CPU reference costs for one tell at this size (bounded samples)
    Xn = np.asarray(opt.space.transform(opt.Xi))
    yn = np.asarray(opt.yi)
    th = opt.gp.theta
    cpu_t = cpu_eval_time(Xn, yn, np.tile(th, (8, 1)), budget=4.0)
    t0 = time.perf_counter()
    O.pvrs_covs(Xn, None, th, Xn[:8] * 0.5 + 0.1, Xn[:10] * 0.3)
    cpu_pvrs = (time.perf_counter() - t0) / 8
    print(json.dumps({"section": "E-cpu", "cpu_ms_per_lml_eval": 1e3 * cpu_t, "cpu_ms_per_pvrs_candidate": 1e3 * cpu_pvrs,
                      "cpu_tell_estimate_s": cpu_t * 1300 + cpu_pvrs * m}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["A", "B", "C", "D", "E"]
    if "A" in which:
        fit_sample(128, 2, 100, 90, 10, "A")
    if "B" in which:
        fit_sample(1024, 8, 64, 490, 10, "B")
    if "C" in which:
        fit_sample(2048, 16, 256, 20, 5, "C")
    if "D" in which:
        config_d()
    if "E" in which:
        config_e()
