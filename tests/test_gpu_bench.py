"""bench.py contract: one JSON line with the driver's keys plus `roofline`, on one GPU and as a 2-rank job
(both ranks on GPU 0 over gloo -- the box has one device) in both sharding modes."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _line(res):
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout[-2000:]  # ONE JSON line and nothing else on stdout
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=1500)
    d = _line(res)
    assert KEYS <= set(d)
    assert d["metric"] == "mcmc_lml_evals_per_s_n2048" and d["n_gpus"] == 1 and d["dtype"] == "f64"
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert abs(d["value"] - 256 * 2 / (d["ms_per_step"] * 2 * 1e-3)) / d["value"] < 1e-9
    # measurement hygiene: the headline is the median of three timed passes, the sampler's run is resident on the device
    tp = d["timed_passes_ms_per_step"]
    assert len(tp) == 3 and sorted(tp)[1] == d["ms_per_step"] and d["resident"] is True and d["sampler"].startswith("device-resident")
    r = d["roofline"]
    assert abs(r["frac_of_spec"] - r["achieved"] / 78.6) < 1e-12
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak_spec"] == 78.6
    # peak = min(datasheet, measured on the box), both stated
    assert r["peak"] == min(r["peak_spec"], r["mfma_peak_measured"]) and 60.0 < r["mfma_peak_measured"] < 90.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    # HBM bytes per launch: collected by two rocprofv3 --pmc child passes of this very run when the profiler is there
    # (else the committed passes); the algorithmic C round trip alone is 0.45 GB per launch
    assert r["traffic_source"].startswith(("live", "profiles/")) and 0.45e9 < r["traffic"] < 2.0e9
    # the first panel group's launches compute the Gram blocks they touch first (config C: d = 16): `achieved` counts those flops
    # with the launches that carry them, `achieved_update_only` leaves them out; the Gram kernel builds block column 0 only
    assert r["gram_generated_in_kernel"] is True and abs(r["gram_generated_gflop_per_factorisation"] - 0.1142784) < 1e-9
    assert 0.93 < r["achieved_update_only"] / r["achieved"] < 0.97 and abs(r["frac_update_only"] - r["achieved_update_only"] / r["peak"]) < 1e-12
    assert d["kernel_ms_per_half_step"]["kbuild"] < 0.3
    assert d["roofline_n4096"]["gram_generated_gflop_per_factorisation"] == 0.0  # (d = 32: not generated)
    bk = r["by_launch_kind"]  # the same kernel's bulk launches against its look-ahead column launches
    assert 0.5 < bk["look_ahead_columns"]["frac_of_peak"] < bk["bulk"]["frac_of_peak"] < 1.0
    assert abs(bk["bulk"]["ms_per_half_step"] + bk["look_ahead_columns"]["ms_per_half_step"] - d["kernel_ms_per_half_step"]["syrk"]) < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    n4 = d["roofline_n4096"]  # the north star's own target: >= 40 % of fp64 peak on the trailing update at n = 4096
    assert n4["lml_finite"] and n4["peak"] == r["peak"] and 0.4 < n4["frac"] < 1.0
    assert d["fit_plus_sample_ms"] > 0 and d["fit_plus_sample_evals"] == 256 * 31
    # the other BASELINE configurations and the strong-scaling shards, in the driver-run line
    sh = d["shard_ms"]
    assert set(sh) == {"128", "64", "32", "16"} and sh["16"] < sh["32"] < sh["64"] < sh["128"]
    assert set(d["shard_projection"]["evals_per_s"]) == {"1", "2", "4", "8"}
    bud = d["shard_projection"]["per_half_step_budget"]  # what a future SCALE line can be checked against
    assert bud["shard_ms"] == sh and set(bud["collective_ms_in_stream_1rank"]) == {"2", "4", "8"}
    assert all(0 < v < 1.0 for v in bud["collective_ms_in_stream_1rank"].values())
    assert all(bud["evals_per_s_with_budget"][k] < d["shard_projection"]["evals_per_s"][k] for k in ("2", "4", "8"))
    lf = d["launch_free"]
    shapes = {k: v for k, v in lf.items() if k.startswith("n")}
    assert set(shapes) == {"n4096_B1", "n2048_B16", "n1024_B32", "n2048_B1", "n1024_B8"} and all(v["bit_identical"] for v in shapes.values())
    assert all(0 < v["launch_free_ms"] < 3 * v["launches_ms"] for v in shapes.values())
    assert lf["timeouts"] == 0 and all(v["calls"] >= 18 and 0 < v["launch_free_frac"] < 1 for v in shapes.values())
    assert "fresh child process" in lf["measured_in"]
    e2e = d["end_to_end"]  # the whole hot path against the same peak: below the trailing update's own fraction
    assert 0.2 < e2e["frac"] < r["frac"] and abs(e2e["tflops"] - e2e["flops_per_eval"] * d["value"] / 1e12) < 1e-9
    assert d["launch_free_calls_in_timed_path"]["timeouts"] == 0
    # SURVEY 8d in the ONE driver line: every BASELINE configuration with its CPU side, every kernel with its roof
    ca = d["config_A"]  # n = 128, W = 100, 100 steps: the CPU loop runs IN FULL (10 100 evaluations), nothing extrapolated
    assert "10100 evaluations" in ca["workload"] and ca["fit_plus_sample_evals"] == 10100 and ca["evals_per_s"] > 1.0e5
    assert ca["cpu_baseline"]["kind"] == "reference" and ca["cpu_baseline"]["cores"] == 1 and "10100" in ca["cpu_baseline"]["sample"]
    assert ca["speedup_vs_cpu"] > 1.0 and ca["cpu_baseline"]["sample_ms"] > ca["sample_ms"]
    # both small configurations run with the sampler's state resident on the device; the host-driven loop is timed beside it
    assert ca["sampler"].startswith("device-resident") and ca["host_driven_ms_per_half_step"] > 0
    cb = d["config_B"]  # as stated: 500 steps, with the wall clock of a whole fit() and the reference's per-walker call beside it
    assert "500 timed MCMC steps" in cb["workload"] and cb["fit_plus_sample_evals"] == 64 * 501
    assert cb["evals_per_s"] > 2.0e4 and 0.1 < cb["acceptance_fraction"] < 0.9 and cb["fit_plus_sample_ms"] > 0
    assert all(r_["evals"] >= 32 for r_ in cb["cpu_baseline"]["runs"].values()) and cb["speedup_vs_cpu"] > 50
    assert 0.05 < cb["end_to_end"]["frac"] < 1.0 and cb["launch_free_timeouts"] == 0
    assert cb["sampler"].startswith("device-resident") and cb["host_driven_ms_per_half_step"] > 0.9 * cb["ms_per_half_step"]
    ce = d["config_E"]
    assert ce["n_iters"] == 50 and ce["pvrs"]["n_final"] == 1024 and ce["ei128"]["n_final"] == 1024
    assert 0 < ce["pvrs"]["median_ms_per_tell"] < 500 and 0 < ce["ei128"]["median_ms_per_tell"] < 500
    assert "EXTRAPOLATED" in ce["cpu_baseline"]["unit"] and "EXTRAPOLATED" in ce["cpu_baseline"]["sample"]
    assert ce["cpu_baseline"]["value"] > 1e3 and ce["pvrs"]["speedup_vs_cpu_extrapolated"] > 50
    rk = {k_["kernel"].split(" ")[0]: k_ for k_ in d["roofline_kernels"]}
    assert {"syrk4_kernel<64>", "trsm4_kernel", "xscale_kernel", "potrf_kernel", "ps_kernel"} <= set(rk)
    assert rk["trsm4_kernel"]["bound"] == "hbm" and rk["trsm4_kernel"]["unit"] == "GB/s" and 0.2 < rk["trsm4_kernel"]["frac"] < 1.0
    assert rk["xscale_kernel"]["bound"].startswith("hbm") and 0.1 < rk["xscale_kernel"]["frac"] < 1.0 and 0 < rk["xscale_kernel"]["valu"]["frac"] < 1
    assert rk["potrf_kernel"]["bound"] == "latency" and 0.1 < rk["potrf_kernel"]["frac"] < 1.0
    assert rk["ps_kernel"]["bound"] == "mfma" and 0.1 < rk["ps_kernel"]["frac"] < 1.0
    for k_ in d["roofline_kernels"]:
        assert {"kernel", "bound", "achieved", "peak", "unit", "frac"} <= set(k_)
    assert bk["bulk"]["bound"] == "mfma" and bk["look_ahead_columns"]["bound"] in ("mfma", "hbm")
    assert (bk["look_ahead_columns"]["bound"] == "mfma") == (bk["look_ahead_columns"]["flop_per_byte"] >= bk["look_ahead_columns"]["ridge_flop_per_byte"])
    assert 0.2 < bk["look_ahead_columns"]["hbm_frac"] < 1.0 and 0.4 < bk["look_ahead_columns"]["frac_of_roof"] <= 1.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    # CPU baselines: the best of 1 / 8 / 16 / 64 BLAS threads (as many of them as the host's pool has), all stated
    for cb_ in (d["cpu_baseline"], d["cpu_baseline_sklearn"], cb["cpu_baseline"]):
        thr = sorted(r_["threads"] for r_ in cb_["runs"].values())
        assert thr[0] == 1 and len(thr) >= 2 and set(thr) <= {1, 2, 4, 8, 16, 64} | {thr[-1]}
        assert cb_["value"] == max(r_["evals_per_s"] for r_ in cb_["runs"].values())
    assert "EXTRAPOLATED" in d["cpu_fit_plus_sample"]["label"]


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver calls it): the parent starts two
    fresh rank processes itself, both on GPU 0 here (one device on the box -> the ranks agree on gloo: RCCL refuses two
    ranks on one device), and relays ONE line: BASELINE config C as stated -- one 256-walker ensemble split over the
    ranks, strong scaling -- with the weak sub-ensemble rate as an extra key."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                            "BGP_DIST_BACKEND")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    d = _line(res)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["parallelism"] == "ensemble_sharded2" and d["config"]["proposals_per_gpu_per_half_step"] == 64
    assert d["config"]["walkers_total"] == 256
    assert abs(d["value"] - 256 * 2 / (d["ms_per_step"] * 2 * 1e-3)) / d["value"] < 1e-9
    assert d["dist_backend"] == "gloo" and d["rank_devices"] == [0, 0] and d["rccl_nranks"] is None
    assert d["weak_chains_evals_per_s"] > 0 and d["gathered_chain_rows"] == 2 * 256
    assert 0 < d["collective_ms_per_half_step"] < 50.0


@pytest.mark.parametrize("shard", ["chains", "ensemble"])
def test_bench_two_ranks_on_one_gpu(shard):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BGP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--shard", shard]
    d = _line(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900))
    assert KEYS <= set(d) and d["n_gpus"] == 2
    if shard == "chains":   # independent 256-walker sub-ensemble per rank: whole-job evaluations = 2 x
        assert d["scaling"] == "weak" and d["gathered_chain_rows"] == 2 * 2 * 256  # ranks x kept steps x walkers
        assert d["strong_ensemble_evals_per_s"] > 0
        assert abs(d["value"] - 2 * 256 * 2 / (d["ms_per_step"] * 2 * 1e-3)) / d["value"] < 1e-9
    else:                   # ONE 256-walker ensemble split over the ranks
        assert d["scaling"] == "strong" and d["config"]["proposals_per_gpu_per_half_step"] == 64
        assert d["weak_chains_evals_per_s"] > 0
        assert abs(d["value"] - 256 * 2 / (d["ms_per_step"] * 2 * 1e-3)) / d["value"] < 1e-9

