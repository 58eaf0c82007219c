"""Exact single-ensemble sharding on the device (SURVEY.md 8(e) option 1): two ranks (gloo rendezvous,
both on GPU 0 -- the GPU box has one device) each evaluate half of every proposal block with the HIP
path and all-gather the log-probabilities; the chain must equal the single-process device chain."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_sharded_fit_equals_single_process(tmp_path):
    import bayes_skopt_amd as bask

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", BGP_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_shard_worker.py"),
           str(tmp_path), "gpu"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]

    rng = np.random.RandomState(0)
    X = rng.uniform(size=(96, 2))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(96)
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1]), random_state=3, device=0, normalize_y=True)
    gp.fit(X, y, n_desired_samples=60, n_burnin=4, n_walkers_per_thread=20, progress=False)
    c0, c1 = np.load(tmp_path / "chain0.npy"), np.load(tmp_path / "chain1.npy")
    np.testing.assert_array_equal(c0, c1)          # all ranks hold the same ensemble
    u0 = np.load(tmp_path / "chain_unsharded0.npy")
    np.testing.assert_array_equal(c0, u0)          # ... and it is the unsharded chain, bit for bit,
    np.testing.assert_array_equal(c0, gp.chain_)   # in the worker and in this process (reproducible device path)
    r0 = json.load(open(tmp_path / "shard0.json"))
    assert np.isclose(r0["lp_sum"], gp.log_marginal_likelihood_value_, rtol=1e-12)


def test_two_rank_sharded_fit_with_a_generic_kernel_tree(tmp_path):
    """The sharded ensemble with a kernel tree that has no canonical device form (host-evaluated kernel matrices, device
    factorisation): every rank evaluates its rows and the finished values are gathered from the host; the chain is the same on
    both ranks and equals the unsharded chain of the same process bit for bit."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", BGP_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_shard_worker.py"),
           str(tmp_path), "gpu_generic"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    c0, c1 = np.load(tmp_path / "chain0.npy"), np.load(tmp_path / "chain1.npy")
    np.testing.assert_array_equal(c0, c1)
    np.testing.assert_array_equal(c0, np.load(tmp_path / "chain_unsharded0.npy"))
    assert c0.shape == (60, 3)  # the two length scales and the WhiteKernel fit() appends


_RCCL_WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
import bayes_skopt_amd as bask
from bayes_skopt_amd import distributed
rank, local_rank, ws = distributed.init_process_group()
assert distributed.backend() == "rccl" and "torch" not in sys.modules, (distributed.backend(), "torch" in sys.modules)
chain = np.arange(12.0).reshape(6, 2) + 100.0 * rank
allc = distributed.gather_chains(chain)
tmax = distributed.max_over_ranks(3.25 + rank)
b = distributed.broadcast_array(np.array([1.5, -2.0, 7.0]))
lp = distributed.shard_log_prob(lambda T: T.sum(axis=1))(np.arange(10.0).reshape(5, 2))
info = distributed.group_info()
distributed.barrier()
big = distributed.gather_chains(np.random.RandomState(1).randn(5120, 18))   # the bench's final gather, again (buffers reused)
distributed.destroy_process_group()
print(json.dumps({"ws": ws, "allc": allc.tolist(), "tmax": tmax, "b": b.tolist(), "lp": lp.tolist(),
                  "big": [big.shape[0], float(big.sum())], "torch": "torch" in sys.modules, "info": info}))
"""


def test_native_rccl_collectives_world_size_one():
    """The multi-GPU exchange goes through libbgp's own RCCL communicator (bgp_comm_*, no PyTorch).  The GPU box
    has one device, so this is a one-rank group (BGP_DIST_FORCE=1): ncclCommInitRank + all-gather / all-reduce(MAX) /
    broadcast really execute on the MI355X before the driver's 8-GPU run does."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), BGP_DIST_FORCE="1",
               BGP_DIST_BACKEND="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, "-c", _RCCL_WORKER % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["ws"] == 1 and d["torch"] is False
    assert d["info"]["backend"] == "rccl" and d["info"]["rccl_nranks"] == 1 and d["info"]["rank_devices"] == [0]
    np.testing.assert_array_equal(np.array(d["allc"]), np.arange(12.0).reshape(6, 2))
    assert d["tmax"] == 3.25 and d["b"] == [1.5, -2.0, 7.0] and d["lp"] == [1.0, 5.0, 9.0, 13.0, 17.0]
    ref = np.random.RandomState(1).randn(5120, 18)
    assert d["big"][0] == 5120 and abs(d["big"][1] - ref.sum()) < 1e-9


def test_bench_line_through_native_rccl_group():
    """bench.py under the launcher with the RCCL group forced at world size 1: barrier, max-over-ranks and the final
    chain gather run through bgp_comm_*."""
    env = dict(os.environ, BGP_DIST_FORCE="1", BGP_DIST_BACKEND="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--no-extras"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    # the bench line is the ONLY thing on standard output: RCCL's start-up banner ("RCCL version : ...") goes to stderr
    out_lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1 and out_lines[0].startswith("{"), res.stdout[-1500:]
    assert "RCCL version" in res.stderr
    d = json.loads(out_lines[0])
    assert d["dist_backend"] == "rccl" and d["n_gpus"] == 1 and d["gathered_chain_rows"] == 2 * 256
    # the forced one-rank group runs the SHARDED ensemble (one rank's share = all 128 proposals): resident on the device, the
    # per-half-step all-gather on the context's stream, its price measured in-stream
    assert d["scaling"] == "strong" and d["config"]["parallelism"] == "ensemble_sharded1" and d["resident"] is True
    assert d["sampler"].startswith("device-resident") and "all-gathered on the context's stream" in d["sampler"]
    assert 0 < d["collective_ms_per_half_step"] < 1.0 and d["collective_note"].startswith("in-stream")
    assert len(d["timed_passes_ms_per_step"]) == 3


_GATHER_WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
import bayes_skopt_amd as bask
from bayes_skopt_amd import distributed
rank, local_rank, ws = distributed.init_process_group()
rng = np.random.RandomState(0)
X = rng.uniform(size=(300, 3)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(300)
kw = dict(n_desired_samples=80, n_burnin=3, n_walkers_per_thread=20, progress=False)
gs = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1, 2]), random_state=3, device=0, normalize_y=True,
                   shard_ensemble=True).fit(X, y, **kw)
calls = []
orig = gs._ctx.lml_wait_allgather
g1 = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1, 2]), random_state=3, device=0, normalize_y=True).fit(X, y, **kw)
# and once more driven from the host, with the collective counted
gs2 = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1, 2]), random_state=3, device=0, normalize_y=True,
                    shard_ensemble=True, resident_sampler=False)
from bayes_skopt_amd import _lib
real = _lib.Context.lml_wait_allgather
def counted(self, comm, per_rank, local_error=0):
    calls.append(per_rank)
    return real(self, comm, per_rank, local_error)
_lib.Context.lml_wait_allgather = counted
gs2.fit(X, y, **kw)
_lib.Context.lml_wait_allgather = real
# the sharded run resident on the device at the launch-free shard's shape (16 matrices of n = 1100), with the reference's default
# progress bar, against the unsharded resident run and against the host-driven sharded run
rng = np.random.RandomState(1)
X2 = rng.uniform(size=(1100, 4)); y2 = np.sin(3.0 * X2.sum(axis=1)) + 0.1 * rng.randn(1100)
kw2 = dict(n_desired_samples=32 * 6, n_burnin=2, n_walkers_per_thread=32)
big = [bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1, 2, 3]), random_state=5, device=0, normalize_y=True,
                     shard_ensemble=sh, resident_sampler=res).fit(X2, y2, **kw2) for sh, res in ((True, True), (False, True), (True, False))]
stats = big[0]._ctx.persist_stats()
distributed.destroy_process_group()
print(json.dumps({"same": bool(np.array_equal(gs.chain_, g1.chain_) and np.array_equal(gs2.chain_, g1.chain_)),
                  "resident": [getattr(g._sampler, "resident_runs", 0) for g in (gs, g1, gs2)],
                  "big_same": bool(np.array_equal(big[0].chain_, big[1].chain_) and np.array_equal(big[0].chain_, big[2].chain_)),
                  "big_resident": [getattr(g._sampler, "resident_runs", 0) for g in big], "big_stats": stats,
                  "calls": len(calls), "per_rank": sorted(set(calls)), "torch": "torch" in sys.modules}))
"""


def test_device_resident_lml_gather_through_rccl():
    """The per-half-step exchange of the exact single-ensemble sharding: bgp_lml_batch_wait_allgather gathers the
    log-likelihoods out of the context's device-resident result vector over RCCL (one-rank group here: the box has one
    GPU).  A sharded fit through it reproduces the unsharded chain bit for bit."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), BGP_DIST_FORCE="1",
               BGP_DIST_BACKEND="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, "-c", _GATHER_WORKER % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["same"] and d["torch"] is False
    # the sharded fit is resident on the device (the all-gather sits on the context's stream between the LML batch and the next
    # step kernel: no host call per half-step); the host-driven form gathers once per half-step
    assert d["resident"] == [1, 1, 0]
    assert d["calls"] == 1 + 2 * 7 and d["per_rank"] == [10, 20]  # initial ensemble (20 rows) + 2 half-steps x 7 steps
    assert d["big_same"] and d["big_resident"] == [1, 1, 0]
    assert d["big_stats"]["calls"] >= 2 * 8 and d["big_stats"]["timeouts"] == 0  # launch-free batches beside the RCCL kernels


_ERRPATH_WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
import bayes_skopt_amd as bask
from bayes_skopt_amd import _lib, distributed
rank, local_rank, ws = distributed.init_process_group()
comm = distributed._state["comm"]
rng = np.random.RandomState(0)
n, d = 1100, 4
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
ctx = _lib.Context(X, y, 1e-10, max_batch=16)
H = np.concatenate([[0.0], np.full(d, np.log(0.4)), [np.log(0.02)]]) + 0.1 * rng.randn(12, d + 2)
ctx.set_persist(0)
ref = ctx.lml(H)
out = {}
# 1. sound batch, launch schedule and launch-free: values == the synchronous call, status words 0
for mode in (0, 1):
    ctx.set_persist(mode)
    assert ctx.lml_submit(H)
    vals, errs = ctx.lml_wait_allgather(comm, 16)
    out["sound%%d" %% mode] = [bool(np.array_equal(vals[0, :12], ref)), errs.tolist(), bool(np.all(np.isnan(vals[0, 12:])))]
# 2. this rank reports a failure of its own: it still takes part, the values are NaN, the word carries the code, and the
#    pending batch is consumed (the context is usable at once)
assert ctx.lml_submit(H)
vals, errs = ctx.lml_wait_allgather(comm, 16, local_error=7)
out["err"] = [errs.tolist(), bool(np.all(np.isnan(vals)))]
out["after_err"] = bool(np.array_equal(ctx.lml(H), ref))
# 3. nothing submitted (a rank without rows): padding only
vals, errs = ctx.lml_wait_allgather(comm, 16)
out["empty"] = [errs.tolist(), bool(np.all(np.isnan(vals)))]
# 4. per_rank too small for the pending batch: reported THROUGH the collective, not by returning before it
assert ctx.lml_submit(H)
vals, errs = ctx.lml_wait_allgather(comm, 8)
out["small"] = errs.tolist()
out["after_small"] = bool(np.array_equal(ctx.lml(H), ref))
# 5. the sharded log-probability: a failure in the priors reaches every rank as ShardedEvaluationError (cause attached)
gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=3, device=0, normalize_y=True,
                   shard_ensemble=True)
def bad_prior(x):
    raise FloatingPointError("prior blew up")
try:
    gp.fit(X[:200], y[:200], n_desired_samples=40, n_burnin=1, n_walkers_per_thread=20, progress=False,
           priors=[bad_prior] * (d + 2))
    out["fit"] = "no error"
except distributed.ShardedEvaluationError as exc:
    out["fit"] = [str(exc), repr(exc.__cause__)]
ctx.close()
distributed.destroy_process_group()
print(json.dumps(out))
"""


def test_collective_error_path_through_rccl():
    """bgp_lml_batch_wait_allgather never returns before the collective: a local failure travels as a status word next to
    NaN values, the pending batch is consumed whatever happens, and a timed-out launch-free factorisation is redone by
    launches with a second gather round (one-rank RCCL group: the box has one GPU)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), BGP_DIST_FORCE="1",
               BGP_DIST_BACKEND="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, "-c", _ERRPATH_WORKER % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["sound0"] == [True, [0], True] and d["sound1"] == [True, [0], True]
    assert d["err"] == [[7], True] and d["after_err"]
    assert d["empty"] == [[0], True]
    assert d["small"] == [1] and d["after_small"]  # BGP_ERR_INVALID
    assert "[0]" in d["fit"][0] and "FloatingPointError" in d["fit"][1]
    # the same with every launch-free wait timing out: the rank's word says "redo", it redoes its batch by launches and the
    # gather goes round again -- same values, one warning
    res = subprocess.run([sys.executable, "-c", _ERRPATH_WORKER % ROOT], env=dict(env, BGP_PS_TIMEOUT_TICKS="200"),
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["sound1"] == [True, [0], True] and res.stderr.count("timed out") >= 1


_SOAK_WORKER = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, %r)
import bayes_skopt_amd as bask
from bayes_skopt_amd import _lib, distributed
rank, local_rank, ws = distributed.init_process_group()
comm = distributed._state["comm"]
rng = np.random.RandomState(0)
n, d, B = 2048, 16, 16     # the per-GPU share of BASELINE config C at N = 8: 16 matrices of n = 2048 (launch-free by the automatic rule)
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
ctx = _lib.Context(X, y, 1e-10, max_batch=B)
base = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
blocks = [base + 1e-2 * rng.randn(B, d + 2) for _ in range(4)]
ctx.set_persist(0)
refs = [ctx.lml(H) for H in blocks]
ctx.set_persist(-1)        # automatic: what a rank of the sharded ensemble runs with
t0 = time.perf_counter()
bad = 0
for it in range(2000):
    H = blocks[it %% 4]
    assert ctx.lml_submit(H)
    vals, errs = ctx.lml_wait_allgather(comm, B)
    bad += int(not np.array_equal(vals[0], refs[it %% 4])) + int(np.any(errs != 0))
stats = ctx.persist_stats()
print(json.dumps({"bad": bad, "stats": [stats["calls"], stats["timeouts"], int(stats["disabled"])], "ms_per_half_step": (time.perf_counter() - t0) / 2000 * 1e3}))
ctx.close()
distributed.destroy_process_group()
"""


def test_launch_free_shard_beside_a_live_rccl_communicator_2000_half_steps():
    """The 16-matrix shard of the sharded ensemble takes the launch-free factorisation by the automatic rule; include/bgp.h tells
    contexts that share their device with collectives to switch it off.  They do not have to: the per-half-step all-gather waits
    for the context's stream through an event and the next batch is submitted after the host has the gathered values, so
    ps_kernel and the RCCL kernel are never on the device together.  2000 half-steps of exactly that shape through a live RCCL
    communicator (one rank: the box has one GPU): every value bit-identical to the launch schedule, every status word 0, zero
    time-outs."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), BGP_DIST_FORCE="1",
               BGP_DIST_BACKEND="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.pop("BGP_PERSIST", None)
    res = subprocess.run([sys.executable, "-c", _SOAK_WORKER % ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    print(d)
    assert d["bad"] == 0
    assert d["stats"][0] == 2000 and d["stats"][1] == 0 and d["stats"][2] == 0, d  # 2000 launch-free calls, no time-out
