#!/usr/bin/env python3
"""bench.py -- the BASELINE.json headline metric on MI355X: MCMC LML-evaluations/s of the BayesGPR
hot path at n=2048 (config C: d=16, Matern-5/2 ARD + White, 256 walkers), plus the roofline of the
dominant kernel (fp64-MFMA trailing update of the blocked Cholesky) and a CPU baseline.

A "step" is one ensemble-MCMC step = two half-steps, each one batched device call that builds,
factorises and scores W/2 = 128 kernel matrices; K steps = 256*K LML evaluations.
Inputs (X, y, walker positions) are resident / tiny; only (128, 18) doubles of proposals go up and
128 log-likelihoods come back per half-step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1, one process per GPU either way: without a launcher's RANK / WORLD_SIZE in the environment `bench.py --gpus N`
starts its N ranks itself (fresh child processes, before this process has imported the package or made any HIP
call) and relays rank 0's line.  The headline is BASELINE config C AS STATED: ONE 256-walker ensemble (what
bask/bayesgpr.py:490-530 runs) whose 128 proposals per half-step are split over the GPUs -- 16 per GPU at N = 8 --
with the log-likelihoods all-gathered device to device over RCCL each half-step (bgp_lml_batch_wait_allgather through
libbgp's own communicator, no PyTorch in the path): `"scaling": "strong"`, chain bit-identical to the one-GPU chain.
The rate of N independent 256-walker sub-ensembles (no collective in the loop, final gather only; `--shard chains`
makes it the headline) is timed behind it and reported as `weak_chains_evals_per_s`.  `rccl_nranks` (ncclCommCount),
`rank_devices` and `dist_backend` say what group really formed.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import signal
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS, N_DIMS, N_WALKERS = 2048, 16, 256
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet fp64 matrix peak (dense); not in MI355X_MICROARCH.md
NB = 128


def synth(n, d, seed):
    """SURVEY.md 8(d) synthetic design."""
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    return X, (y - y.mean()) / y.std()


def trailing_flops_per_launch(n, nb=NB):
    """F_trail(n, nb) per matrix, split per launch j: nb * m_j * (m_j + 1),
    m_j = n - j*nb (lower-triangular syrk, 2 flop per MAC)."""
    return [nb * (n - j * nb) * (n - j * nb + 1) for j in range(1, n // nb)]


def lml_flops(n, d):
    """Algorithmic flops of ONE log-marginal-likelihood evaluation (sklearn/_gpr.py:579-613): the Gram build on the lower
    triangle ((3 d + 14) flop per pair), the Cholesky factorisation n^3 / 3 and the forward substitution n^2."""
    return n * n / 2.0 * (3 * d + 14) + n ** 3 / 3.0 + float(n) * n


def trailing_flops_split(n, nb=NB):
    """The same flops by KIND of launch of the multi-panel schedule (csrc/bgp_chol.hip: P = 4 block columns per group from
    12 block columns, else 2): look-ahead COLUMN launches (block column c = k+j+1 of a group that started at k gets the
    panels k .. k+j, K = nb (j+1), on the 128 m_c - 8128 lower elements of that column) and BULK launches (everything
    to the right of the group with K = nb P).  Returns (F_column, F_bulk) per matrix; their sum is sum(F_trail)."""
    nblk = n // nb
    P = 4 if nblk >= 12 else 2
    total = float(sum(trailing_flops_per_launch(n, nb)))
    col = 0.0
    k = 0
    while k < nblk:
        np_ = min(P, nblk - k)
        for j in range(np_ - 1):
            c = k + j + 1
            m_c = n - c * nb
            col += 2.0 * nb * (j + 1) * (nb * m_c - nb * (nb - 1) / 2.0)
        k += np_
    return col, total - col


def gram_generated_flops(n, d, nb=NB):
    """Gram flops the trailing update's generating launches carry when the library builds only block column 0 with the Gram
    kernel (csrc/bgp_s4.h S4GenF, bgp_lml_gen_stats): (3 d + 14) flop per pair of the lower triangle outside block
    column 0, split as trailing_flops_split splits the update: (look-ahead columns of the first panel group, its bulk update).
    Pairs of block column j: nb (n - j nb) - nb^2 / 2 (their sum over all columns is the n^2 / 2 of lml_flops)."""
    nblk = n // nb
    np0 = min(4 if nblk >= 12 else 2, nblk)
    per = lambda j: nb * (n - j * nb) - nb * nb / 2.0  # noqa: E731
    col = sum(per(j) for j in range(1, np0))
    bulk = sum(per(j) for j in range(np0, nblk))
    return col * (3 * d + 14), bulk * (3 * d + 14)


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E ~ 8 TB/s (6.29 TB/s measured streaming copy)
HBM_COPY_MEASURED_GBS = 6290.0
PIVOT_CHAIN_NS = 81.0      # measured latency of one pivot of the in-register 16 x 16 micro-Cholesky (tools/potrf_bench, PF_TRACE)


def hot_path_bytes(n, d, nb=NB):
    """Algorithmic HBM bytes per matrix and half-step of the launch schedule's HBM-side kernels:
    Gram build: every lower 128 x 128 tile written once; panel solve: the sub-diagonal block column read and written once per
    block column; look-ahead column launches: block column c read and written once + its K = 128 (j+1) panel rows read once."""
    nblk = n // nb
    kbuild = nblk * (nblk + 1) // 2 * nb * nb * 8.0
    trsm = sum(2.0 * 8.0 * nb * (n - (k + 1) * nb) for k in range(nblk - 1))
    P = 4 if nblk >= 12 else 2
    col, k = 0.0, 0
    while k < nblk:
        np_ = min(P, nblk - k)
        for j in range(np_ - 1):
            c = k + j + 1
            m_c = n - c * nb
            col += 16.0 * (nb * m_c - nb * (nb - 1) / 2.0) + 8.0 * m_c * nb * (j + 1)
        k += np_
    return {"kbuild": kbuild, "trsm": trsm, "syrk_columns": col}


def _PROFILER_ENV(k):
    return k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_", "ROCTX_")) or k in ("LD_PRELOAD", "HSA_TOOLS_LIB")


def being_profiled():
    """True when this process runs under rocprofv3 / rocprof (their tool library is preloaded or configured)."""
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("HSA_TOOLS_LIB", "")
    return "rocprof" in pre or any(k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_")) for k in os.environ)


def _kill_group(proc, grace=3.0):
    """End a child started with start_new_session=True together with everything it started (exact process group)."""
    try:
        os.killpg(proc.pid, signal.SIGTERM)
    except (ProcessLookupError, PermissionError):
        return
    try:
        proc.wait(timeout=grace)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        proc.wait()


def live_pmc_traffic(kernel_sub="syrk4_kernel", timeout_s=240):
    """HBM bytes per launch of the trailing-update kernel, collected NOW: two child runs of this script's hot path
    (`--steps 1 --warmup 1 --no-extras`, one stream) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
    `... --pmc WRITE_SIZE` (separate passes, MI355X_MICROARCH.md section HBM), FETCH_SIZE x 2 (gfx950 correction for
    wide coalesced reads) + WRITE_SIZE, counters in KiB.  None when rocprofv3 is missing or a pass fails."""
    import glob
    import shutil
    import sqlite3
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe or being_profiled():
        # (under a profiler this process already carries its preloaded tool library: a nested `rocprofv3` -- a
        # `#!/usr/bin/env python3` script -- would be an exec from a GPU-initialised process)
        return None
    per_launch = {}
    env = {k: v for k, v in os.environ.items() if not _PROFILER_ENV(k)}
    env.update(BGP_STREAMS="1", TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="bgp_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-extras"]
            # own session: on a timeout the whole group goes (rocprofv3 AND the bench it started), so nothing of this
            # pass can linger on the GPU under the measurements that follow
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rcode = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                _kill_group(proc)
                return None
            dbs = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
            if rcode != 0 or not dbs:
                return None
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                               "group by kernel_name", (counter,)).fetchall()
            cnt = sum(c for name, c, _ in rows if kernel_sub in name)
            tot = sum(v for name, _, v in rows if kernel_sub in name)
            if cnt == 0:
                return None
            per_launch[counter] = (tot / cnt, cnt)
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    fetch_kb, write_kb = per_launch["FETCH_SIZE"][0], per_launch["WRITE_SIZE"][0]
    return {"traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0, "fetch_size_kb_raw": fetch_kb,
            "write_size_kb_raw": write_kb, "launches_averaged": per_launch["FETCH_SIZE"][1]}


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info

        # the BLAS pools only: an OpenMP pool (scikit-learn's) reports every core of the host, and raising OpenBLAS above the
        # thread count it was initialised with (OPENBLAS_NUM_THREADS in the environment, as tests/conftest.py sets) crashes dpotrf
        return max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        return os.cpu_count() or 1


def _thread_sweep():
    """BLAS thread counts the CPU baselines are timed at: 1, 8, 16 and 64 where the pool has them (on a 256-core host one thread
    beat all 64 at n = 2048, and BASELINE.md's own probe had 8 best for the n = 2048 Cholesky): the baseline is the BEST of them."""
    top = _blas_threads()
    return sorted({t for t in (1, 8, 16, 64) if t <= top} | ({top} if top < 8 else set()))


def _label(nthreads):
    return "1_thread" if nthreads == 1 else "%d_threads" % nthreads


def _timed_evals(fn, thetas, budget_s, min_evals=3):
    """Sequential evaluations (one walker at a time, as emcee's map would) until the budget is used up, at least
    `min_evals`; returns the per-evaluation times."""
    fn(thetas[0])  # warm-up
    times = []
    t_start = time.perf_counter()
    for th in thetas:
        t0 = time.perf_counter()
        fn(th)
        times.append(time.perf_counter() - t0)
        if len(times) >= min_evals and time.perf_counter() - t_start > budget_s:
            break
    return times


def cpu_baseline(X, y, thetas, budget_s=4.0):
    """Oracle (numpy/scipy restatement of sklearn's log_marginal_likelihood) on the host cores, a bounded sample of
    the same workload, at 1 / 8 / 16 / 64 BLAS threads; `value` = the BEST of them from the median
    evaluation time, all of them stated."""
    from threadpoolctl import threadpool_limits

    from oracle import gp_oracle as O

    ad = np.full(len(y), 1e-10)
    runs = {}
    for nthreads in _thread_sweep():
        with threadpool_limits(limits=nthreads):
            t = _timed_evals(lambda th: O.lml(X, y, ad, th), thetas, budget_s)
        runs[_label(nthreads)] = {"threads": int(nthreads), "evals": len(t), "median_ms_per_eval": float(np.median(t) * 1e3),
                                  "evals_per_s": float(1.0 / np.median(t))}
    best = max(runs, key=lambda k: runs[k]["evals_per_s"])
    return {
        "value": runs[best]["evals_per_s"],
        "unit": "LML-evals/s",
        "cores": runs[best]["threads"],
        "kind": "port",
        "host_cpu_count": os.cpu_count(),
        "runs": runs,
        "sample": f"sequential LML evaluations (n={len(y)}, d={X.shape[1]}) of walker positions from the same start "
        "ball with oracle/gp_oracle.py (numpy + scipy LAPACK): "
        + ", ".join("%d at %d BLAS thread(s)" % (r["evals"], r["threads"]) for r in runs.values())
        + "; median time per evaluation; value = the fastest setting",
    }


def _sklearn_gpr(X, y):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import ConstantKernel, Matern, WhiteKernel

    d = X.shape[1]
    k = ConstantKernel(1.0, (0.1, 2.0)) * Matern(length_scale=[0.3] * d, length_scale_bounds=(0.2, 0.5), nu=2.5) \
        + WhiteKernel(0.01)
    return GaussianProcessRegressor(kernel=k, optimizer=None, alpha=1e-10).fit(X, y)


def cpu_baseline_sklearn(X, y, thetas, budget_s=3.0):
    """The call the reference itself makes per walker (bask/bayesgpr.py:374): scikit-learn's
    GaussianProcessRegressor.log_marginal_likelihood(theta) -- third-party code present in the image on both
    sides, timed on the same bounded sample next to the oracle restatement (they agree to 1e-12)."""
    from threadpoolctl import threadpool_limits

    gpr = _sklearn_gpr(X, y)
    vals = []

    def f(th):
        vals.append(gpr.log_marginal_likelihood(th))

    runs = {}
    for nthreads in _thread_sweep():
        del vals[:]
        with threadpool_limits(limits=nthreads):
            t = _timed_evals(f, thetas, budget_s)
        runs[_label(nthreads)] = {"threads": int(nthreads), "evals": len(t), "median_ms_per_eval": float(np.median(t) * 1e3),
                                  "evals_per_s": float(1.0 / np.median(t))}
    best = max(runs, key=lambda k: runs[k]["evals_per_s"])
    return {"value": runs[best]["evals_per_s"], "unit": "LML-evals/s", "cores": runs[best]["threads"],
            "kind": "sklearn 1.7 GaussianProcessRegressor.log_marginal_likelihood", "runs": runs}, vals[1:]


def cpu_fit_plus_sample(X, y, priors, theta0, n_walkers_full, steps_full, budget_walkers=36, budget_steps=1, threads=None):
    """TIMED host run of the reference's sampling loop at this size: the package's host ensemble sampler (emcee's
    stretch move, one proposal block per half-step) driving scikit-learn's log_marginal_likelihood one walker at a
    time -- what BayesGPR.sample does on the CPU (bask/bayesgpr.py:510-530, :351-379: a sequential per-walker map,
    whatever the host's core count).  Bounded: `budget_walkers` walkers (the smallest ensemble emcee accepts, 2p) x
    `budget_steps` steps, run at ONE BLAS thread and at the thread count the per-evaluation sweep found best (`threads`); the
    better setting is the baseline (as in cpu_baseline) and the full-size figure is its time per evaluation x the evaluations
    of the full run: an EXTRAPOLATION, and labelled so."""
    from threadpoolctl import threadpool_limits

    from bayes_skopt_amd.sampler import EnsembleSampler

    gpr = _sklearn_gpr(X, y)
    n_eval = [0]

    def log_prob(Theta, priors=None):
        out = np.empty(len(Theta))
        for i, th in enumerate(Theta):
            lp = sum(float(pr(t)) for pr, t in zip(priors, th))
            out[i] = lp + gpr.log_marginal_likelihood(th) if np.isfinite(lp) else -np.inf
            n_eval[0] += 1
        return out

    p = len(theta0)
    W = max(budget_walkers, 2 * p)
    evals_full = n_walkers_full * (steps_full + 1)
    runs = {}
    for nthreads in sorted(set(threads or [1, _blas_threads()])):
        label = _label(nthreads)
        rng = np.random.RandomState(0)
        pos = theta0 + 1e-2 * rng.randn(W, p)
        smp = EnsembleSampler(W, p, log_prob, kwargs=dict(priors=priors))
        smp.random_state = np.random.RandomState(1).get_state()
        n_eval[0] = 0
        with threadpool_limits(limits=nthreads):
            t0 = time.perf_counter()
            smp.run_mcmc(pos, budget_steps)
            dt = time.perf_counter() - t0
        runs[label] = {"threads": int(nthreads), "timed_ms": dt * 1e3, "timed_evals": int(n_eval[0]),
                       "ms_per_eval": dt * 1e3 / max(n_eval[0], 1)}
    best = min(runs, key=lambda k: runs[k]["ms_per_eval"])
    return {
        "runs": runs,
        "best": best,
        "timed_config": f"{W} walkers x {budget_steps} step(s) (+ initial ensemble), sklearn log_marginal_likelihood per "
        f"walker, sequential over the walkers as the reference runs them, on a {os.cpu_count()}-core host",
        "ms_per_eval": runs[best]["ms_per_eval"],
        "label": "EXTRAPOLATED from the timed evaluations above, not run in full",
        "extrapolated_full_ms": runs[best]["ms_per_eval"] * evals_full,
        "extrapolated_full_evals": int(evals_full),
        "extrapolation": f"timed ms per evaluation (best of the BLAS thread counts above) x {evals_full} evaluations = "
        f"{n_walkers_full} walkers x ({steps_full} steps + initial ensemble); the MAP start of fit() (a few dozen more "
        "evaluations with gradients) is not included",
    }


def config_d_roofline(bask_lib, device, peak_tflops):
    """BASELINE config D (n=4096, d=32, blocked fp64 Cholesky, MFMA trailing update): B = 8 matrices per batch, every
    launch on one stream with HIP events around it (as for the config C roofline).  Reports the trailing update's
    algorithmic TFLOP/s against the fp64 MFMA peak -- the north star's ">= 40 % at n = 4096" target."""
    n, d, B = 4096, 32, 8
    X, y = synth(n, d, seed=0)
    ctx = bask_lib.Context(X, y, 1e-10, max_batch=B, device=device)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * np.random.RandomState(3).randn(B, d + 2)
    ctx.set_streams(1)
    ctx.lml(H)  # warm-up
    ctx.set_timing(True)
    reps, syrk_ms, launches, total_ms = 5, 0.0, 0, 0.0
    gen0 = ctx.gen_stats()["batches"]
    for _ in range(reps):
        lml = ctx.lml(H)
        tm = ctx.last_timing()
        syrk_ms += tm["syrk"]["ms"]
        launches += tm["syrk"]["launches"]
        total_ms += tm["device_total_ms"]
    ctx.set_timing(False)
    generated = ctx.gen_stats()["batches"] - gen0 == reps
    t0 = time.perf_counter()
    ctx.lml(H[:1])
    b1_ms = (time.perf_counter() - t0) * 1e3
    ctx.close()
    flops = float(sum(trailing_flops_per_launch(n))) * B * reps
    gen_fl = float(sum(gram_generated_flops(n, d))) * B * reps if generated else 0.0
    achieved = (flops + gen_fl) / (syrk_ms * 1e-3) / 1e12
    return {
        "workload": f"n={n}, d={d}, {B} matrices per batch (BASELINE config D)",
        "kernel": "syrk4_kernel<64> (trailing update, four-panel groups: K = 128..512"
        + ("; the first group's launches generate the Gram blocks they touch first)" if generated else ")"),
        "achieved": achieved, "peak": peak_tflops, "unit": "TFLOP/s", "frac": achieved / peak_tflops,
        "achieved_update_only": flops / (syrk_ms * 1e-3) / 1e12, "frac_update_only": flops / (syrk_ms * 1e-3) / 1e12 / peak_tflops,
        "gram_generated_gflop_per_factorisation": gen_fl / (B * reps) / 1e9,
        "avg_launch_ms": syrk_ms / max(launches, 1), "launches_per_factorisation": launches // reps,
        "ms_per_batch_of_8": total_ms / reps, "ms_single_matrix_wall": b1_ms,
        "algorithmic_flops_per_factorisation": float(sum(trailing_flops_per_launch(n))),
        "lml_finite": bool(np.all(np.isfinite(lml))),
    }


def small_batch_shards(bask_lib, X, y, pos_H, device, sizes=(128, 64, 32, 16), reps=12):
    """Wall time of ONE half-step's device call at n = 2048 for the per-GPU share of the 128 proposals when ONE
    256-walker ensemble is split over 1 / 2 / 4 / 8 GPUs (128 / 64 / 32 / 16 matrices): what strong scaling of BASELINE
    config C is bounded by, measured on this one GPU."""
    ctx = bask_lib.Context(X, y, 1e-10, max_batch=max(sizes), device=device)
    out = {}
    for m in sizes:
        H = pos_H[:m]
        for _ in range(3):
            ctx.lml(H)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.lml(H)
            ts.append(time.perf_counter() - t0)
        out[str(m)] = float(np.median(ts) * 1e3)
    ctx.close()
    return out


# (the first three are the round-3 review's shapes; 2048 x 1 and 1024 x 8 are shapes the chain pairs take: DESIGN.md section 4)
LF_SHAPES = ((4096, 32, 1), (2048, 16, 16), (1024, 8, 32), (2048, 16, 1), (1024, 8, 8))


def launch_free(bask_lib, device, peak=None, shapes=LF_SHAPES, reps=15):
    """The launch-free factorisation of small batches (automatic at these sizes) next to the launch
    schedule: wall ms per LML call, the whole call's algorithmic TFLOP/s (Gram build + factorisation + solve; `frac` of the
    fp64 MFMA peak), whether the log-likelihoods are the same bits, and the path's own bookkeeping (calls, time-outs).  A
    FRESH context per shape and mode, launch-free measured first: the order no longer favours either side."""
    out = {}
    for n, d, B in shapes:
        X, y = synth(n, d, seed=0)
        H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * np.random.RandomState(5).randn(B, d + 2)
        rec, vals = {}, {}
        fl = lml_flops(n, d) * B
        for tag, mode in (("launch_free", 1), ("launches", 0)):
            ctx = bask_lib.Context(X, y, 1e-10, max_batch=B, device=device)
            ctx.set_persist(mode)
            for _ in range(3):
                vals[tag] = ctx.lml(H)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                ctx.lml(H)
                ts.append(time.perf_counter() - t0)
            ms = float(np.median(ts) * 1e3)
            rec[tag + "_ms"] = ms
            rec[tag + "_tflops"] = fl / (ms * 1e-3) / 1e12
            if peak:
                rec[tag + "_frac"] = rec[tag + "_tflops"] / peak
            if mode == 1:
                st = ctx.persist_stats()
                rec["calls"], rec["timeouts"] = st["calls"], st["timeouts"]
            ctx.close()
        rec["bit_identical"] = bool(np.array_equal(vals["launches"], vals["launch_free"]))
        out[f"n{n}_B{B}"] = rec
    out["timeouts"] = int(sum(v["timeouts"] for v in out.values()))
    return out


def launch_free_fresh_process(device, peak, timeout_s=240):
    """`launch_free` in a CHILD process started for it (python bench.py --launch-free-only): late in this long process the
    same calls read 7-10 % slow or fast depending on where the allocator put their buffers (tools/archive/lf_place_probe.py); a
    number the driver records should not carry that."""
    if being_profiled():
        return None
    env = {k: v for k, v in os.environ.items() if not _PROFILER_ENV(k)}
    cmd = [sys.executable, os.path.abspath(__file__), "--launch-free-only", "--device", str(device), "--peak", repr(peak)]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        _kill_group(proc)
        return None
    for ln in reversed(out.splitlines()):
        if ln.startswith("{"):
            rec = json.loads(ln)
            rec["measured_in"] = "a fresh child process (python bench.py --launch-free-only)"
            return rec
    return None


def _host_sampler_cpu(X, y, priors, theta0, W, steps, seed=1):
    """The reference's sampling loop on the host: the package's ensemble sampler driving scikit-learn's
    log_marginal_likelihood one walker at a time (bask/bayesgpr.py:510-530, :351-379), at one BLAS thread.  Returns
    (wall seconds, evaluations)."""
    from threadpoolctl import threadpool_limits

    from bayes_skopt_amd.sampler import EnsembleSampler

    gpr = _sklearn_gpr(X, y)
    n_eval = [0]

    def log_prob(Theta, priors=None):
        out = np.empty(len(Theta))
        for i, th in enumerate(Theta):
            lp = sum(float(pr(t)) for pr, t in zip(priors, th))
            out[i] = lp + gpr.log_marginal_likelihood(th) if np.isfinite(lp) else -np.inf
            n_eval[0] += 1
        return out

    p = len(theta0)
    pos = theta0 + 1e-2 * np.random.RandomState(0).randn(W, p)
    smp = EnsembleSampler(W, p, log_prob, kwargs=dict(priors=priors))
    smp.random_state = np.random.RandomState(seed).get_state()
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        smp.run_mcmc(pos, steps)
        dt = time.perf_counter() - t0
    return dt, int(n_eval[0])


def _device_sampler(bask, device, n, d, W, steps, warm=2):
    """The device side of a BASELINE configuration's MCMC: BayesGPR in the state right after the MAP fit, the default priors,
    the reference's start ball; `steps` timed ensemble steps behind `warm` untimed ones.  Returns (gp, sampler, state, seconds)."""
    from bayes_skopt_amd.bayesgpr import _AsyncLogProb
    from bayes_skopt_amd.kernels import WhiteKernel

    X, y = synth(n, d, seed=0)
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0, device=device,
                       max_batch=W // 2)
    gp.kernel_ = gp.kernel + WhiteKernel(noise_level=0.01)
    gp.noise_ = 0.01
    gp.X_train_, gp.y_train_ = X, y
    gp.y_train_mean_, gp.y_train_std_ = np.zeros(1), 1
    gp._ensure_context(batch_hint=W // 2)
    priors = bask.guess_priors(gp.kernel_)
    theta0 = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
    pos = theta0 + 1e-2 * gp.random_state.randn(W, d + 2)
    smp = bask.sampler.EnsembleSampler(W, d + 2, _AsyncLogProb(gp), kwargs=dict(priors=priors))
    smp.random_state = np.random.RandomState(1).get_state()
    st = smp.run_mcmc(pos, warm)
    t0 = time.perf_counter()
    st = smp.run_mcmc(st.coords, steps, log_prob0=st.log_prob, skip_initial_state_check=True)
    dt = time.perf_counter() - t0
    return gp, smp, st, dt, (X, y, priors, theta0)


def config_a(bask, device, with_cpu=True):
    """BASELINE config A as stated: n = 128, d = 2, Matern-5/2, W = 100 walkers (the reference's default), 100 MCMC steps =
    10 100 log-likelihood evaluations.  Device: ONE fused launch per half-step (proposal, Gram generation, factorisation, LML and
    accept test in the walker's workgroup: the device-resident sampler).  CPU: the same 100 steps of the host loop (scikit-learn's log_marginal_likelihood per walker, one
    BLAS thread), run IN FULL -- no extrapolation."""
    n, d, W, steps = 128, 2, 100, 100
    gp, smp, st, dt, (X, y, priors, theta0) = _device_sampler(bask, device, n, d, W, steps)
    resident_runs = int(getattr(smp, "resident_runs", 0))
    gp.resident_sampler = False  # (the same steps driven from the host, continuing the chain)
    th0 = time.perf_counter()
    smp.run_mcmc(st.coords, steps, log_prob0=st.log_prob, skip_initial_state_check=True)
    host_ms = (time.perf_counter() - th0) / (2 * steps) * 1e3
    gp._ctx.close()
    tf0 = time.perf_counter()
    gp2 = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0, device=device)
    gp2.fit(X, y, n_desired_samples=W * steps, n_burnin=0, n_walkers_per_thread=W, progress=False)
    fit_ms = (time.perf_counter() - tf0) * 1e3
    out = {"workload": f"n={n}, d={d}, {W} walkers x {steps} steps ({W * (steps + 1)} evaluations incl. the start ensemble)",
           "evals_per_s": W * steps / dt, "ms_per_half_step": dt / (2 * steps) * 1e3, "sample_ms": dt * 1e3,
           "sampler": ("device-resident, ONE launch per half-step (each workgroup proposes, builds, factorises and accepts its own walker)"
                       if resident_runs else "host-driven (one LML batch call per half-step)"),
           "host_driven_ms_per_half_step": host_ms,
           "fit_plus_sample_ms": fit_ms, "fit_plus_sample_evals": int(gp2._sampler.n_log_prob_evals),
           "acceptance_fraction": float(np.mean(smp.acceptance_fraction))}
    del gp2
    if with_cpu:
        cdt, cev = _host_sampler_cpu(X, y, priors, theta0, W, steps)
        out["cpu_baseline"] = {"value": cev / cdt, "unit": "LML-evals/s", "cores": 1, "kind": "reference",
                               "sample": f"the WHOLE configuration on the host: {W} walkers x {steps} steps = {cev} sequential "
                               "sklearn log_marginal_likelihood calls + priors, host ensemble sampler, one BLAS thread",
                               "sample_ms": cdt * 1e3, "ms_per_eval": cdt * 1e3 / cev}
        out["speedup_vs_cpu"] = (W * steps / dt) / (cev / cdt)
    return out


def config_b(bask, device, steps=500, with_cpu=True, peak_tflops=None):
    """BASELINE config B as stated (n = 1024, d = 8, Matern-5/2, 64 walkers x 500 steps = 32 064 evaluations): MCMC
    LML-evaluations/s over all 500 steps, ms per half-step (32 proposals), the wall clock of a whole BayesGPR.fit() at that
    size, the per-kernel split of one half-step on the launch schedule (HIP events, one stream), and the CPU baseline: the
    reference's per-walker call timed on >= 32 walker positions."""
    n, d, W = 1024, 8, 64
    gp, smp, st, dt, (X, y, priors, theta0) = _device_sampler(bask, device, n, d, W, steps, warm=5)
    ps = gp._ctx.persist_stats()
    resident_runs = int(getattr(smp, "resident_runs", 0))
    # the same steps driven from the host (one LML batch call per half-step: upload, launch, synchronise, download, numpy
    # bookkeeping), continuing the chain: what the device-resident run saves per half-step
    gp.resident_sampler = False
    th0 = time.perf_counter()
    smp.run_mcmc(st.coords, 100, log_prob0=st.log_prob, skip_initial_state_check=True)
    host_ms = (time.perf_counter() - th0) / 200 * 1e3
    gp.resident_sampler = True
    H = gp._canonical(st.coords[: W // 2])
    gp._ctx.set_streams(1)
    gp._ctx.set_timing(True)
    split = {k: 0.0 for k in ("kbuild", "potrf", "trsm", "syrk")}
    dev = 0.0
    for _ in range(5):
        gp._ctx.lml(H)
        tm = gp._ctx.last_timing()
        for k in split:
            split[k] += tm[k]["ms"] / 5
        dev += tm["device_total_ms"] / 5
    gp._ctx.set_timing(False)
    gp._ctx.close()
    tf0 = time.perf_counter()
    gp2 = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0, device=device)
    gp2.fit(X, y, n_desired_samples=W * steps, n_burnin=0, n_walkers_per_thread=W, progress=False)
    fit_ms = (time.perf_counter() - tf0) * 1e3
    evals_fit = int(gp2._sampler.n_log_prob_evals)
    del gp2
    rate = W * steps / dt
    out = {"workload": f"n={n}, d={d}, {W} walkers, {steps} timed MCMC steps (32 proposals per half-step)",
           "evals_per_s": rate, "ms_per_half_step": dt / (2 * steps) * 1e3,
           "sampler": ("device-resident (bgp_mcmc_begin / _steps / _end: no transfer or synchronisation between half-steps)"
                       if resident_runs else "host-driven (one LML batch call per half-step)"),
           "host_driven_ms_per_half_step": host_ms,
           "fit_plus_sample_ms": fit_ms, "fit_plus_sample_evals": evals_fit,
           "fit_plus_sample_config": f"BayesGPR.fit: MAP start (L-BFGS-B on the device LML + gradient) + {W} walkers x {steps} steps",
           "launch_free_calls": ps["calls"], "launch_free_timeouts": ps["timeouts"],
           "kernel_ms_per_half_step_launch_schedule": split, "device_ms_per_half_step_instrumented": dev,
           "acceptance_fraction": float(np.mean(smp.acceptance_fraction))}
    if peak_tflops:
        out["end_to_end"] = {"flops_per_eval": lml_flops(n, d), "tflops": lml_flops(n, d) * rate / 1e12,
                             "frac": lml_flops(n, d) * rate / 1e12 / peak_tflops}
    if with_cpu:
        from threadpoolctl import threadpool_limits

        gpr = _sklearn_gpr(X, y)
        runs = {}
        for nthreads in _thread_sweep():
            with threadpool_limits(limits=nthreads):
                t = _timed_evals(lambda th: gpr.log_marginal_likelihood(th), st.coords, budget_s=2.0, min_evals=32)
            runs[_label(nthreads)] = {"threads": int(nthreads), "evals": len(t), "median_ms_per_eval": float(np.median(t) * 1e3),
                                      "evals_per_s": float(1.0 / np.median(t))}
        best = max(runs, key=lambda k: runs[k]["evals_per_s"])
        out["cpu_baseline"] = {"value": runs[best]["evals_per_s"], "unit": "LML-evals/s", "cores": runs[best]["threads"],
                               "kind": "reference", "runs": runs,
                               "sample": f">= 32 sequential sklearn log_marginal_likelihood calls (n={n}, d={d}) on walker "
                               "positions of the timed chain, at 1 / 8 / 16 / 64 BLAS threads; value = the fastest setting"}
        out["speedup_vs_cpu"] = rate / out["cpu_baseline"]["value"]
        out["cpu_fit_plus_sample_ms_extrapolated"] = evals_fit / out["cpu_baseline"]["value"] * 1e3
    return out


def config_e_cpu(n=1000, d=8, m=10000, n_thompson=10, mcmc_evals=1300, svd_m=1500):
    """The CPU side of ONE config-E PVRS tell, timed PIECEWISE on bounded samples and EXTRAPOLATED (a full CPU tell is
    minutes): (i) the MCMC's log-likelihood calls (sklearn, n ~ 1000), (ii) the PVRS loop's bordered (n+1) Cholesky + solve per
    candidate (bask/acquisition.py:328-338), (iii) the Thompson draw's SVD multivariate normal over the m candidates
    (sklearn/_gpr.py:522-526), timed at `svd_m` points and scaled by (m / svd_m)^3."""
    from scipy.linalg import cho_solve, cholesky
    from threadpoolctl import threadpool_limits

    X, y = synth(n, d, seed=0)
    gpr = _sklearn_gpr(X, y)
    k = gpr.kernel_
    rng = np.random.RandomState(0)
    th = k.theta
    out = {}
    best = None
    for nthreads in _thread_sweep():
        label = _label(nthreads)
        with threadpool_limits(limits=nthreads):
            t = _timed_evals(lambda t_: gpr.log_marginal_likelihood(t_), th + 0.01 * rng.randn(24, len(th)), budget_s=1.5, min_evals=12)
            lml_ms = float(np.median(t) * 1e3)
            T = rng.uniform(size=(n_thompson, d))
            cand = rng.uniform(size=(12, d))
            ts = []
            for x in cand:
                t0 = time.perf_counter()
                Xa = np.vstack([X, x[None, :]])
                L = cholesky(k(Xa), lower=True)
                kt = k(Xa, T)
                float(np.sum(kt * cho_solve((L, True), kt)))
                ts.append(time.perf_counter() - t0)
            pvrs_ms = float(np.median(ts[2:]) * 1e3)
            Xq = rng.uniform(size=(svd_m, d))
            mean, cov = gpr.predict(Xq, return_cov=True)
            t0 = time.perf_counter()
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                np.random.RandomState(1).multivariate_normal(mean, cov, n_thompson)
            svd_ms = (time.perf_counter() - t0) * 1e3
        tell_ms = lml_ms * mcmc_evals + pvrs_ms * m + svd_ms * (m / svd_m) ** 3
        out[label] = {"threads": int(nthreads), "lml_ms_per_eval": lml_ms, "pvrs_ms_per_candidate": pvrs_ms,
                      "svd_mvn_ms_at_m%d" % svd_m: svd_ms, "extrapolated_ms_per_tell": tell_ms}
        if best is None or tell_ms < out[best]["extrapolated_ms_per_tell"]:
            best = label
    return {"value": out[best]["extrapolated_ms_per_tell"], "unit": "ms per PVRS tell (EXTRAPOLATED)", "cores": out[best]["threads"],
            "kind": "reference", "runs": out,
            "sample": f"timed pieces of one tell at n={n}: >= 12 sklearn LML calls x {mcmc_evals} (100 walkers x 13 steps), 10 bordered "
            f"(n+1)-Cholesky + solve iterations of the PVRS loop x {m} candidates, one numpy SVD multivariate normal at m={svd_m} "
            f"x ({m}/{svd_m})^3; EXTRAPOLATED, not run in full (a CPU tell is minutes)"}


def config_e(bask, device, n_iters=50, n0=974, m=10000, d=8):
    """BASELINE config E: the Optimizer.tell loop, 50 iterations, PVRS over a 10 000-candidate grid with 128
    hyper-posterior samples, n growing 975 -> 1024 (bask/optimizer.py:228-380, bask/acquisition.py:316-339); and the
    same 50 tells with EI averaged over 128 hyper-posterior samples x 10 000-point predicts.  Wall time per tell
    (the first tell of each loop also runs the MAP fit and is reported apart)."""
    out = {"n_iters": n_iters, "candidates": m, "d": d}
    for tag, acq, kw in (("pvrs", "pvrs", dict(gp_samples=128, gp_burnin=10, n_samples=0)),
                         ("ei128", "ei", dict(gp_samples=200, gp_burnin=10, n_samples=128))):
        rng = np.random.RandomState(0)

        def f(x):
            return float(np.sin(3 * np.sum(x)) + 0.1 * rng.randn())

        opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * d, n_points=m, n_initial_points=10, init_strategy="r2",
                             acq_func=acq, random_state=0, gp_kwargs=dict(device=device))
        X0 = rng.uniform(size=(n0, d)).tolist()
        opt.tell(X0, [f(x) for x in X0], fit=False)
        times = []
        for _ in range(n_iters):
            x = opt.ask() if opt._next_x is not None else rng.uniform(size=d).tolist()
            t0 = time.perf_counter()
            opt.tell(x, f(x), **kw)
            times.append((time.perf_counter() - t0) * 1e3)
        rest = np.array(times[1:])
        out[tag] = {"first_tell_ms_incl_fit": times[0], "median_ms_per_tell": float(np.median(rest)),
                    "p90_ms_per_tell": float(np.percentile(rest, 90)), "total_s": float(np.sum(times) / 1e3),
                    "n_final": len(opt.Xi), "tell_kwargs": kw}
        del opt
    return out


def self_spawn(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent has not
    imported the package, loaded libbgp or made any HIP call -- nothing is ever re-executed from a process that
    touched the GPU), relay rank 0's single JSON line, exit with the worst child return code."""
    import socket
    import tempfile

    n = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                BGP_COMM_JOB="bench_%d_%d" % (os.getpid(), time.time_ns()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out0 = tempfile.TemporaryFile(mode="w+")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else sys.stderr, start_new_session=True))
    deadline = time.monotonic() + float(os.environ.get("BGP_BENCH_TIMEOUT", "3000"))
    worst = 0
    try:
        while any(p.poll() is None for p in procs):
            failed = [p for p in procs if p.poll() not in (None, 0)]
            if failed or time.monotonic() > deadline:
                time.sleep(5.0 if failed else 0.0)  # (let the others report their own error first)
                for p in procs:
                    if p.poll() is None:
                        _kill_group(p)
                worst = worst or (failed[0].returncode if failed else 124)
                break
            time.sleep(0.05)
    except BaseException:
        for p in procs:
            if p.poll() is None:
                _kill_group(p)
        raise
    for p in procs:
        rc = p.wait()
        worst = worst or rc
    out0.seek(0)
    lines = [ln for ln in out0.read().splitlines() if ln.strip()]
    if lines:
        print(lines[-1], flush=True)
    elif worst == 0:
        worst = 1
    sys.exit(worst if 0 <= worst < 256 else 1)


def setup_config_c(bask, device, seed_gp, W=N_WALKERS):
    """BayesGPR in the state right after the MAP fit of BayesGPR.fit (bask/bayesgpr.py:602-607) at config C, the
    default priors and the reference's start ball (:506-509)."""
    from bayes_skopt_amd.kernels import WhiteKernel

    n, d = N_POINTS, N_DIMS
    X, y = synth(n, d, seed=0)
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=seed_gp, device=device,
                       max_batch=W // 2)
    gp.kernel_ = gp.kernel + WhiteKernel(noise_level=0.01)
    gp.noise_ = 0.01
    gp.X_train_ = X
    gp.y_train_ = y
    gp.y_train_mean_, gp.y_train_std_ = np.zeros(1), 1
    gp._ensure_context(batch_hint=W // 2)
    priors = bask.guess_priors(gp.kernel_)
    theta0 = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
    pos = theta0 + 1e-2 * gp.random_state.randn(W, d + 2)
    return gp, X, y, priors, theta0, pos


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the config C hot path (no fit(), no configs B / D / E, no CPU baselines): what the "
                    "rocprofv3 passes of tools/profile_round.sh run, so that their per-kernel numbers are config C's alone")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not collect roofline.traffic with two rocprofv3 --pmc child passes (~25 s); use the "
                    "newest committed profiles/rNN[_vK]_pmc_traffic.json instead")
    ap.add_argument("--shard", choices=("ensemble", "chains"), default="ensemble",
                    help="N > 1 only.  ensemble (default; BASELINE config C as stated, the reference's semantics): ONE "
                    "256-walker ensemble, each half-step's 128 proposals split over the GPUs + a device-to-device "
                    "all-gather of 128 doubles, strong scaling (SURVEY 8e option 1); chains: an independent 256-walker "
                    "sub-ensemble per GPU, weak scaling, no collective in the loop")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="form the process group, report it (backend, ranks RCCL counts, device of every rank) and exit "
                    "without device work: the launch path alone (the CPU tests run it over gloo)")
    ap.add_argument("--launch-free-only", action="store_true", help="(internal) the launch_free section alone, as one JSON line")
    ap.add_argument("--device", type=int, default=0, help="(internal) device of --launch-free-only")
    ap.add_argument("--peak", type=float, default=0.0, help="(internal) fp64 MFMA peak the fractions are quoted against")
    args = ap.parse_args()
    if args.launch_free_only:
        import bayes_skopt_amd  # noqa: F401
        from bayes_skopt_amd import _lib as lib_only

        print(json.dumps(launch_free(lib_only, args.device, args.peak or None)), flush=True)
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and not ("RANK" in os.environ and "WORLD_SIZE" in os.environ):
        return self_spawn(args, sys.argv[1:])  # (before the package, libbgp or HIP are touched in this process)

    import bayes_skopt_amd as bask
    from bayes_skopt_amd import _lib, distributed
    from bayes_skopt_amd.bayesgpr import _AsyncLogProb, _ShardedLogProb

    ndev = _lib.device_count()
    if ndev < 1 and not args.rendezvous_only:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    rank, local_rank, ws = distributed.world()
    if ws != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher's WORLD_SIZE is {ws}: they must agree (or start "
                         f"`python bench.py --gpus {args.gpus}` without a launcher: it spawns its own ranks)")
    device = local_rank % ndev if ndev > 0 else None
    distributed.init_process_group(device=device)
    info = distributed.group_info(device)
    if args.rendezvous_only:
        seen = distributed.gather_chains(np.full((1, 1), float(rank)))[:, 0].tolist()
        tmax = distributed.max_over_ranks(float(rank))
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": ws, "dist_backend": info["backend"],
                              "rccl_nranks": info["rccl_nranks"], "rank_devices": info["rank_devices"],
                              "ranks_seen": seen, "max_rank": tmax}), flush=True)
        if distributed.backend() is not None:
            distributed.barrier()
            distributed.destroy_process_group()
        return

    n, d, W = N_POINTS, N_DIMS, N_WALKERS
    # (a group forced at world size 1 -- BGP_DIST_FORCE=1, the one-GPU test of the RCCL path -- runs the sharded ensemble too)
    grouped = distributed.backend() is not None
    ensemble = args.shard == "ensemble" and grouped
    seed_rank = 0 if (ensemble or ws == 1) else rank  # one shared ensemble needs the same RNG streams on every rank
    gp, X, y, priors, theta0, pos = setup_config_c(bask, device, distributed.rank_seed(0, seed_rank))

    def make_sampler(shared):
        lp = _ShardedLogProb(gp) if shared else _AsyncLogProb(gp)
        s = bask.sampler.EnsembleSampler(W, d + 2, lp, kwargs=dict(priors=priors))
        s.random_state = np.random.RandomState(distributed.rank_seed(1, 0 if shared else seed_rank)).get_state()
        return s

    def sync():
        _lib.device_synchronize(device)

    def timed(sampler, pos, lp, steps):
        distributed.barrier()
        sync()
        t0 = time.perf_counter()
        state = sampler.run_mcmc(pos, steps, log_prob0=lp, skip_initial_state_check=True)
        sync()
        distributed.barrier()
        return distributed.max_over_ranks(time.perf_counter() - t0), state

    sampler = make_sampler(ensemble)
    state = sampler.run_mcmc(pos, max(args.warmup, 1))  # also evaluates the initial ensemble
    # THREE timed passes of exactly K steps each (barrier + device synchronisation on both sides, MAX over ranks); the headline
    # is their median -- the boxes of this pool differ by ~5 %, and one 0.3 s pass carries whatever the box was doing
    passes = []
    for _ in range(3):
        dt_i, state = timed(sampler, state.coords, state.log_prob, args.steps)
        passes.append(dt_i)
    dt = float(np.median(passes))
    resident_runs = int(getattr(sampler, "resident_runs", 0))
    pos, lp = state.coords, state.log_prob

    # final posterior-sample gather (RCCL over xGMI when N > 1; the shared ensemble's chain is on every rank already)
    tg = time.perf_counter()
    chain_local = sampler.get_chain(flat=True, discard=max(args.warmup, 1) + 2 * args.steps)  # (the last timed pass's steps)
    chain_all = chain_local if ensemble else distributed.gather_chains(chain_local)
    gather_ms = (time.perf_counter() - tg) * 1e3

    # N > 1: the other sharding, timed the same way behind the headline (extra key, never `value`)
    other = None
    if ws > 1:
        if ensemble:  # independent sub-ensembles: own seeds, own start ball, no collective in the loop
            s_o = bask.sampler.EnsembleSampler(W, d + 2, _AsyncLogProb(gp), kwargs=dict(priors=priors))
            s_o.random_state = np.random.RandomState(distributed.rank_seed(1, rank)).get_state()
            pos_o = theta0 + 1e-2 * np.random.RandomState(distributed.rank_seed(0, rank)).randn(W, d + 2)
        else:  # (headline = chains): ONE shared ensemble on rank 0's streams
            s_o = make_sampler(True)
            pos_o = distributed.broadcast_array(pos)
        st_o = s_o.run_mcmc(pos_o, 1)
        dt_o, _ = timed(s_o, st_o.coords, st_o.log_prob, args.steps)
        other = W * args.steps * (ws if ensemble else 1) / dt_o

    # instrumented pass: HIP events around every launch, same work.  The timed pass above runs the product default
    # (two walker-group streams at >= 64 matrices: the groups' launches overlap, so per-launch durations there are not
    # those of a kernel running alone); for the per-kernel numbers and the roofline every launch goes to ONE stream,
    # like the rocprofv3 runs under profiles/ (BGP_STREAMS=1).
    gp._ctx.set_streams(1)
    gp._ctx.set_timing(True)
    gp.resident_sampler = False  # (per-launch timing synchronises inside every batch: the host-driven loop, by choice)
    acc = {k: [0.0, 0] for k in ("kbuild", "potrf", "trsm", "syrk", "syrk_columns")}
    dev_total = 0.0
    n_calls = 0
    orig = gp._ctx.lml

    def timed_lml(H, return_status=False):
        nonlocal dev_total, n_calls
        out = orig(H, return_status)
        tm = gp._ctx.last_timing()
        for k in acc:
            acc[k][0] += tm[k]["ms"]
            acc[k][1] += tm[k]["launches"]
        dev_total += tm["device_total_ms"]
        n_calls += 1
        return out

    gen_before = gp._ctx.gen_stats()
    gp._ctx.lml = timed_lml
    t1 = time.perf_counter()
    sampler.run_mcmc(pos, args.steps, log_prob0=lp, skip_initial_state_check=True)
    dt_instr = time.perf_counter() - t1
    gp._ctx.lml = orig
    gp._ctx.set_timing(False)
    gp.resident_sampler = True

    B = W // 2
    if ensemble:  # this rank's share of every proposal block
        lo, hi = distributed.shard_rows(B, rank, ws)
        B = hi - lo
    fl = trailing_flops_per_launch(n)
    syrk_ms, syrk_launches = acc["syrk"]
    flops_per_call = float(sum(fl)) * B
    # the library builds only block column 0 with the Gram kernel and generates every other block in the accumulators of the first
    # panel group's updates (bgp_lml_gen_stats): those launches then carry the Gram flops of their blocks as well -- fp64 VALU work
    # on the SAME pipe as the fp64 MFMAs (tools/dp_share_probe.hip: the two do not overlap on gfx950; one 78.6 TF peak for both)
    gen_after = gp._ctx.gen_stats()
    n_streams_timed = 1
    generated = n_calls > 0 and gen_after["batches"] - gen_before["batches"] >= n_calls * n_streams_timed
    gen_col_fl, gen_bulk_fl = gram_generated_flops(n, d) if generated else (0.0, 0.0)
    update_only = flops_per_call * n_calls / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
    achieved = (flops_per_call + (gen_col_fl + gen_bulk_fl) * B) * n_calls / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
    # fp64 MFMA peak: min(datasheet, on-box micro-benchmark of back-to-back v_mfma_f64_16x16x4_f64), both stated.  Run
    # right behind the instrumented pass: the chip is power-limited under fp64 MFMA load, so numerator and denominator
    # should see the same clocks
    mfma_measured = None
    if not args.no_extras:  # (the rocprofv3 passes run --no-extras: their kernel tables hold the hot path only)
        try:
            mfma_measured = float(_lib.bench_mfma_f64(device))
        except Exception:
            mfma_measured = None
    traffic, traffic_source = None, None
    if rank == 0 and ws == 1 and not args.no_extras and not args.no_live_pmc:
        live = live_pmc_traffic()
        if live:
            traffic = live["traffic_bytes_per_launch"]
            traffic_source = ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes of this run "
                              "(%d launches; FETCH_SIZE %.0f KiB raw x 2 + WRITE_SIZE %.0f KiB)"
                              % (live["launches_averaged"], live["fetch_size_kb_raw"], live["write_size_kb_raw"]))
    if traffic is None:
        import glob
        import re

        # the NEWEST committed PMC passes of tools/profile_round.sh (config C's: rNN_pmc_traffic.json / rNN_vK_pmc_traffic.json)
        names = [os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))]
        names = [f for f in names if re.fullmatch(r"r\d+(_v\d+)?_pmc_traffic\.json", f)]
        for name in sorted(names, key=lambda f: [int(v) for v in re.findall(r"\d+", f)], reverse=True):
            try:
                traffic = json.load(open(os.path.join(ROOT, "profiles", name)))["traffic_bytes_per_launch"]
                traffic_source = f"profiles/{name} (committed passes of tools/profile_round.sh)"
                break
            except Exception:
                pass
    peak = min(FP64_MFMA_PEAK_TFLOPS, mfma_measured) if mfma_measured else FP64_MFMA_PEAK_TFLOPS
    # where the fraction is lost: the same kernel's bulk launches (K = 512 at this size) against its look-ahead column
    # launches (K = 128 / 256 / 384 on one 128-wide block column)
    f_col, f_bulk = trailing_flops_split(n)
    col_ms, col_launches = acc["syrk_columns"]
    by_kind = None
    if col_ms > 0 and syrk_ms > col_ms:
        by_kind = {
            "bulk": {"tflops": (f_bulk + gen_bulk_fl) * B * n_calls / ((syrk_ms - col_ms) * 1e-3) / 1e12,
                     "tflops_update_only": f_bulk * B * n_calls / ((syrk_ms - col_ms) * 1e-3) / 1e12,
                     "ms_per_half_step": (syrk_ms - col_ms) / n_calls, "launches_per_half_step": (syrk_launches - col_launches) / n_calls},
            "look_ahead_columns": {"tflops": (f_col + gen_col_fl) * B * n_calls / (col_ms * 1e-3) / 1e12,
                                   "tflops_update_only": f_col * B * n_calls / (col_ms * 1e-3) / 1e12,
                                   "ms_per_half_step": col_ms / n_calls, "launches_per_half_step": col_launches / n_calls},
        }
        for v in by_kind.values():
            v["frac_of_peak"] = v["tflops"] / peak
        by_kind["bulk"]["bound"] = "mfma"
        # the K = 128 / 256 / 384 launches on ONE block column sit at the ridge: 16 B per element of the column read + written plus
        # the panel rows once against 2 K flops per element -- 10.7 / 21 / 32 flop per byte, the ridge is peak / 8 TB/s = 9.6
        cb = hot_path_bytes(n, d)["syrk_columns"] * B * n_calls
        ck = by_kind["look_ahead_columns"]
        ck["algorithmic_gb_per_half_step"] = cb / n_calls / 1e9
        ck["tb_per_s"] = cb / (col_ms * 1e-3) / 1e12
        ck["hbm_frac"] = ck["tb_per_s"] * 1e3 / HBM_PEAK_GBS
        ck["flop_per_byte"] = f_col * B * n_calls / cb
        ck["roof_tflops"] = min(peak, ck["flop_per_byte"] * HBM_PEAK_GBS / 1e3)
        ck["frac_of_roof"] = ck["tflops"] / ck["roof_tflops"]
        ck["ridge_flop_per_byte"] = peak / (HBM_PEAK_GBS / 1e3)
        # (above the ridge the MFMA roof is the lower one: these launches are NOT HBM-bound on algorithmic traffic -- they move
        # hbm_frac of the HBM peak -- they are short: 12 launches of 0.15 ms, a C round trip per 128 .. 384 columns of k)
        ck["bound"] = "mfma" if ck["flop_per_byte"] >= ck["ridge_flop_per_byte"] else "hbm"
    roofline = {
        "bound": "mfma",
        "kernel": "syrk4_kernel<64> (blocked-Cholesky trailing update, four-panel groups K = 128..512, LDS-DMA ring, fp64 "
        "v_mfma_f64_16x16x4_f64" + ("; the first group's launches generate the Gram blocks they touch first in their accumulators)"
                                    if generated else ")"),
        "achieved": achieved,
        "achieved_update_only": update_only,
        "frac_update_only": update_only / peak,
        "gram_generated_in_kernel": bool(generated),
        "gram_generated_gflop_per_factorisation": (gen_col_fl + gen_bulk_fl) / 1e9,
        "peak": peak,
        "peak_spec": FP64_MFMA_PEAK_TFLOPS,
        "mfma_peak_measured": mfma_measured,
        "unit": "TFLOP/s",
        "frac": achieved / peak,
        "frac_of_spec": achieved / FP64_MFMA_PEAK_TFLOPS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "avg_launch_ms": syrk_ms / max(syrk_launches, 1),
        "launches": syrk_launches,
        "algorithmic_flops_per_factorisation": float(sum(fl)),
        "by_launch_kind": by_kind,
        "note": "measured with all launches on one stream (kernel alone on the GPU); the timed pass overlaps two "
        "walker-group streams. algorithmic flops = sum_j nb*m_j*(m_j+1) per matrix x the matrices of a launch (SURVEY "
        "8d) + -- when gram_generated_in_kernel -- the (3d+14) flop per pair of the Gram blocks the first panel group's launches "
        "compute instead of loading (every block outside block column 0; fp64 VALU on the pipe the fp64 MFMAs use); "
        "achieved_update_only leaves those out; peak = min(datasheet fp64 matrix peak 78.6 TF, bgp_bench_mfma_f64 measured on this box) -- "
        "MI355X_MICROARCH.md has no fp64 row; traffic = HBM bytes per launch (rocprofv3 FETCH_SIZE x2 gfx950 correction "
        "+ WRITE_SIZE, separate passes; see traffic_source)",
    }

    # every kernel of the half-step against the roof that bounds it (per half-step of this rank: B matrices)
    hb = hot_path_bytes(n, d)
    kms = {k: v[0] / max(n_calls, 1) for k, v in acc.items()}
    roofline_kernels = [dict(kernel="syrk4_kernel<64> (trailing update)", bound="mfma", achieved=achieved, peak=peak, unit="TFLOP/s",
                             frac=achieved / peak, ms_per_half_step=kms["syrk"])]
    if kms["trsm"] > 0:
        gbs = hb["trsm"] * B / (kms["trsm"] * 1e-3) / 1e9
        roofline_kernels.append(dict(kernel="trsm4_kernel (panel solve + fused forward substitution)", bound="hbm", achieved=gbs,
                                     peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, frac_of_measured_copy=gbs / HBM_COPY_MEASURED_GBS,
                                     algorithmic_gb_per_half_step=hb["trsm"] * B / 1e9, ms_per_half_step=kms["trsm"]))
    if kms["kbuild"] > 0:
        # (generated: the Gram kernel writes block column 0 only -- nblk tiles of the nblk (nblk + 1) / 2)
        kshare = (2.0 / (n // NB + 1)) if generated else 1.0
        gbs = hb["kbuild"] * kshare * B / (kms["kbuild"] * 1e-3) / 1e9
        fk = (n * n / 2.0 * (3 * d + 14) - (gen_col_fl + gen_bulk_fl)) * B
        roofline_kernels.append(dict(kernel="xscale_kernel + kbuild2_kernel (Gram build)", bound="hbm (written once) / fp64 VALU",
                                     achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                                     algorithmic_gb_per_half_step=hb["kbuild"] * kshare * B / 1e9, ms_per_half_step=kms["kbuild"],
                                     block_column_0_only=bool(generated),
                                     valu={"algorithmic_tflops": fk / (kms["kbuild"] * 1e-3) / 1e12, "peak_tflops": FP64_MFMA_PEAK_TFLOPS,
                                           "frac": fk / (kms["kbuild"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                           "note": "(3d+14) flop per pair (SURVEY 8d) against the fp64 vector peak; the kernel executes 78 "
                                           "VALU lane-instructions per pair: VALUBusy 84 % (profiles/r03_kbuild_counters.txt)"}))
    if kms["potrf"] > 0 and acc["potrf"][1] > 0:
        us = acc["potrf"][0] / acc["potrf"][1] * 1e3
        floor = NB * PIVOT_CHAIN_NS * 1e-3
        roofline_kernels.append(dict(kernel="potrf_kernel (128 x 128 diagonal block, one workgroup per matrix)", bound="latency",
                                     achieved=us, peak=floor, unit="us per launch (lower is better)", frac=floor / us,
                                     ms_per_half_step=kms["potrf"],
                                     note="peak = 128 dependent pivots x %.0f ns (the in-register pivot chain alone, tools/potrf_bench); "
                                     "frac = that floor / the measured launch" % PIVOT_CHAIN_NS))

    evals = W * args.steps * (1 if (ensemble or ws == 1) else ws)
    value = evals / dt
    line = {
        "metric": "mcmc_lml_evals_per_s_n2048",
        "value": value,
        "unit": "LML-evals/s",
        "n_gpus": ws,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "timed_passes_ms_per_step": [t / args.steps * 1e3 for t in passes],
        "value_is": "the median of three timed passes of exactly `steps` steps each",
        "sampler": ("device-resident (bgp_mcmc_begin_ex / _steps / _end: no transfer or host synchronisation between half-steps"
                    + ("; the ranks' log-likelihoods all-gathered on the context's stream)" if ensemble else ")"))
                   if resident_runs else "host-driven (one LML batch call" + (" + one all-gather" if ensemble else "") + " per half-step)",
        "resident": bool(resident_runs),
        "env_defaults": _lib.env_defaults(),
        "higher_is_better": True,
        "scaling": "strong" if ensemble else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE config C: BayesGPR hyper-posterior MCMC, n=2048, d=16, c*Matern52(ARD)+White, "
            + ("ONE 256-walker ensemble, the 128 proposals of each half-step split over the GPUs and their "
               "log-likelihoods all-gathered device to device over RCCL, " if ensemble else
               "256 walkers per GPU (128 batched kernel-build+Cholesky+LML per half-step), ")
            + "start ball of bask/bayesgpr.py:506-509, default priors",
            "n": n,
            "d": d,
            "walkers_total": W if (ensemble or ws == 1) else W * ws,
            "proposals_per_gpu_per_half_step": B,
            "parallelism": f"ensemble_sharded{ws}" if ensemble else f"chains{ws}",
        },
        "roofline": roofline,
        "roofline_kernels": roofline_kernels,
        "end_to_end": {
            "what": "the WHOLE hot path against the same peak: algorithmic flops of one log-likelihood evaluation (Gram build "
            "(3d+14) flop per pair of the lower triangle + n^3/3 + n^2) x value / peak -- Gram build, diagonal blocks, panel "
            "solves and host bookkeeping included, not the trailing update alone",
            "flops_per_eval": lml_flops(n, d), "tflops": lml_flops(n, d) * value / 1e12,
            "frac": lml_flops(n, d) * value / 1e12 / (peak * ws),
        },
        "kernel_ms_per_half_step": {k: v[0] / max(n_calls, 1) for k, v in acc.items() if k != "syrk_columns"},
        "device_ms_per_half_step": dev_total / max(n_calls, 1),
        "instrumented_ms_per_step": dt_instr / args.steps * 1e3,
        "gather_ms": gather_ms,
        "dist_backend": info["backend"],
        "rccl_nranks": info["rccl_nranks"],
        "rank_devices": info["rank_devices"],
        "gathered_chain_rows": int(chain_all.shape[0]),
        "acceptance_fraction": float(np.mean(sampler.acceptance_fraction)),
    }
    ps = gp._ctx.persist_stats()
    line["launch_free_calls_in_timed_path"] = {"calls": ps["calls"], "timeouts": ps["timeouts"]}
    if grouped:
        # the path's one exchange, alone: all-gather of the per-rank share of 128 doubles (+ status word), barrier-aligned
        per = -(-(W // 2) // ws)
        comm = distributed.communicator()
        if comm is not None:
            # where the resident sampler pays it: pack kernel + RCCL all-gather back to back ON THE CONTEXT'S STREAM between two HIP
            # events (bgp_comm_bench_lml_gather), no host in the loop
            distributed.barrier()
            line["collective_ms_per_half_step"] = distributed.max_over_ranks(comm.bench_lml_gather(gp._ctx, per, 200))
            line["collective_note"] = ("in-stream: 200 rounds of the pack kernel + ncclAllGather of %d doubles per rank on the context's "
                                       "stream between two HIP events, MAX over ranks" % (per + 1))
        else:
            buf = np.zeros(per + 1)
            for _ in range(20):
                distributed._allgather(buf)
            distributed.barrier()
            t0 = time.perf_counter()
            for _ in range(200):
                distributed._allgather(buf)
            line["collective_ms_per_half_step"] = distributed.max_over_ranks((time.perf_counter() - t0) / 200 * 1e3)
            line["collective_note"] = "host-staged all-gather of the same size through the gloo group (the host-driven exchange)"
    if other is not None:
        line["weak_chains_evals_per_s" if ensemble else "strong_ensemble_evals_per_s"] = other
        line["other_sharding_note"] = ("N independent 256-walker sub-ensembles, no collective in the loop (weak scaling; reads "
                                       "~N x by construction)" if ensemble else
                                       "ONE 256-walker ensemble split over the GPUs (strong scaling, BASELINE config C as stated)")
    if args.no_extras:
        args.no_cpu_baseline = True
    if rank == 0 and ws == 1 and not args.no_extras:
        # strong-scaling ceiling of config C, measured on this one GPU: one half-step's device call for the per-GPU
        # share of the 128 proposals at N = 1 / 2 / 4 / 8
        try:
            sh = small_batch_shards(_lib, X, y, gp._canonical(pos[: W // 2]), device)
            line["shard_ms"] = sh
            line["shard_projection"] = {
                "what": "PROJECTION, not a multi-GPU measurement: 256 evaluations per step / (2 x shard_ms[128 / N]) -- "
                "the rate ONE 256-walker ensemble split over N GPUs would reach if the per-half-step all-gather and the "
                "host bookkeeping were free",
                "evals_per_s": {str(N): 256.0 / (2.0 * sh[str(128 // N)] * 1e-3) for N in (1, 2, 4, 8)},
            }
            # ... and the per-half-step BUDGET a future SCALE line can be checked against: the shard's device call (above, wall
            # incl. its launch and synchronisation), the resident run's step kernel, and the in-stream exchange measured on a
            # ONE-rank RCCL group of this GPU (pack kernel + ncclAllGather + their kernel boundaries: the device-side floor of the
            # exchange; the xGMI hop of a real N-rank gather comes on top and is NOT in it)
            budget = {"shard_ms": sh, "step_kernel_ms": 0.0105, "step_kernel_source": "profiles/r05_resident_timeline.txt",
                      "host_ms_per_half_step": 0.0, "host_note": "resident run: the host draws the plan AHEAD of the device"}
            try:
                if _lib.comm_available():
                    c1 = _lib.Comm(device, 0, 1, _lib.comm_unique_id())
                    budget["collective_ms_in_stream_1rank"] = {str(N): c1.bench_lml_gather(gp._ctx, 128 // N, 200) for N in (2, 4, 8)}
                    c1.close()
                    budget["evals_per_s_with_budget"] = {
                        str(N): 256.0 / (2.0 * (sh[str(128 // N)] + budget["step_kernel_ms"]
                                                + budget["collective_ms_in_stream_1rank"][str(N)]) * 1e-3) for N in (2, 4, 8)}
            except Exception as exc:
                budget["collective_ms_in_stream_1rank"] = {"error": repr(exc)}
            line["shard_projection"]["per_half_step_budget"] = budget
        except Exception as exc:
            line["shard_ms"] = {"error": repr(exc)}
        # the other half of BASELINE.json's metric: wall clock of a whole BayesGPR.fit() (MAP start by L-BFGS-B on
        # the device LML + gradient, then 256 walkers x 25 steps after 5 burn-in steps) at the same size
        gp2 = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0, device=device)
        tf0 = time.perf_counter()
        gp2.fit(X, y, n_desired_samples=W * 25, n_burnin=5, n_walkers_per_thread=W, progress=False)
        line["fit_plus_sample_ms"] = (time.perf_counter() - tf0) * 1e3
        line["fit_plus_sample_evals"] = int(gp2._sampler.n_log_prob_evals)
        line["fit_plus_sample_config"] = f"BayesGPR.fit: MAP start (L-BFGS-B on the device LML + gradient) + {W} walkers x 30 steps"
        del gp2
        line["roofline_n4096"] = config_d_roofline(_lib, device, peak)
        try:
            line["launch_free"] = launch_free_fresh_process(device, peak) or launch_free(_lib, device, peak)
        except Exception as exc:
            line["launch_free"] = {"error": repr(exc)}
        with_cpu = not args.no_cpu_baseline
        for key, fn in (("config_A", lambda: config_a(bask, device, with_cpu)),
                        ("config_B", lambda: config_b(bask, device, 500, with_cpu, peak)),
                        ("config_E", lambda: config_e(bask, device))):
            try:
                line[key] = fn()
            except Exception as exc:  # reported, never fatal for the bench line
                line[key] = {"error": repr(exc)}
        if with_cpu and isinstance(line.get("config_E"), dict) and "pvrs" in line["config_E"]:
            try:
                ce = config_e_cpu()
                line["config_E"]["cpu_baseline"] = ce
                line["config_E"]["pvrs"]["speedup_vs_cpu_extrapolated"] = ce["value"] / line["config_E"]["pvrs"]["median_ms_per_tell"]
            except Exception as exc:
                line["config_E"]["cpu_baseline"] = {"error": repr(exc)}
        # the launch-free kernel against the MFMA roof, at config B's own half-step shape (whole call: Gram build + factorisation)
        lf = line.get("launch_free") or {}
        if isinstance(lf.get("n1024_B32"), dict) and "launch_free_tflops" in lf["n1024_B32"]:
            r = lf["n1024_B32"]
            line["roofline_kernels"].append(dict(kernel="ps_kernel (launch-free factorisation, n = 1024 x 32: config B's half-step; whole call)",
                                                 bound="mfma", achieved=r["launch_free_tflops"], peak=peak, unit="TFLOP/s",
                                                 frac=r["launch_free_tflops"] / peak, ms_per_call=r["launch_free_ms"]))
    if rank == 0:
        if not args.no_cpu_baseline and ws == 1:  # (the CPU legs run at N = 1 only: the N > 1 runs time the GPUs)
            line["cpu_baseline"] = cpu_baseline(X, y, pos[:64])
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
            try:  # the CPU side of BASELINE.json's fit+sample metric, timed (bounded) instead of estimated
                best_thr = line["cpu_baseline"]["cores"]
                cf = cpu_fit_plus_sample(X, y, priors, theta0, W, 30, threads=[1, best_thr])
                line["cpu_fit_plus_sample_ms"] = cf["extrapolated_full_ms"]
                line["cpu_fit_plus_sample"] = cf
                if "fit_plus_sample_ms" in line:
                    line["fit_plus_sample_speedup_vs_cpu"] = cf["extrapolated_full_ms"] / line["fit_plus_sample_ms"]
            except Exception as exc:
                line["cpu_fit_plus_sample"] = {"error": repr(exc)}
            try:  # the reference's own per-walker call, and a live parity check of the device path against it
                sk, sk_vals = cpu_baseline_sklearn(X, y, pos[:32])
                dev_vals = gp._ctx.lml(gp._canonical(pos[: len(sk_vals)]))
                sk["max_rel_diff_device_vs_sklearn"] = float(np.max(np.abs(dev_vals - np.array(sk_vals)) / np.abs(sk_vals)))
                line["cpu_baseline_sklearn"] = sk
            except Exception as exc:  # reported, never fatal for the bench line
                line["cpu_baseline_sklearn"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    if grouped:
        distributed.barrier()
        distributed.destroy_process_group()


if __name__ == "__main__":
    main()
