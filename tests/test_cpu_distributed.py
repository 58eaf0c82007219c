"""N>1 path on CPU: world_size-2 gloo job (torch.distributed.run), independent sub-ensembles per rank,
final all-gather of the chains, max-over-ranks timing; exact single-ensemble sharding; the ncclUniqueId rendezvous
(job-unique files, every rank's verdict before anyone enters the collective); bench.py starting its own ranks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gloo_chain_gather(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"),
           str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "[Gloo]" not in res.stdout  # (the transport's connection chatter is kept off the job's standard output)
    r = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    for k in range(2):
        assert r[k]["ws"] == 2 and r[k]["rank"] == k
        assert r[k]["local_shape"] == [20 * 12, 3]
        assert r[k]["all_shape"] == [2 * 20 * 12, 3]  # rank-major concatenation on every rank
        assert r[k]["own_slice_ok"]
        assert r[k]["tmax"] == 2.0
    assert r[0]["checksum_all"] == r[1]["checksum_all"]
    assert np.isclose(r[0]["checksum_local"] + r[1]["checksum_local"], r[0]["checksum_all"])
    assert r[0]["checksum_local"] != r[1]["checksum_local"]  # independent sub-ensembles (rank seeds differ)


def _run_two_ranks(worker, args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", BGP_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", worker)] + list(args)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-2000:]


def test_two_rank_exact_ensemble_sharding_equals_single_process(tmp_path):
    """SURVEY 8(e) option 1: rows of every proposal block split over the ranks + all-gather of the
    log-probabilities == the single-process chain, bit for bit, on every rank."""
    import bayes_skopt_amd as bask

    _run_two_ranks("_dist_shard_worker.py", [str(tmp_path), "toy"])
    p, W, steps = 3, 14, 25
    mu = np.array([1.0, -2.0, 0.5])

    def log_prob(Xb):
        lp = -0.5 * ((Xb - mu) ** 2).sum(axis=1)
        lp[Xb[:, 0] > 1.5] = -np.inf
        return lp

    sampler = bask.sampler.EnsembleSampler(W, p, log_prob)
    sampler.random_state = np.random.RandomState(5).get_state()
    sampler.run_mcmc(mu + 1e-2 * np.random.RandomState(4).randn(W, p), steps)
    ref = sampler.get_chain(flat=True)
    for k in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"chain{k}.npy"), ref)
    r = [json.load(open(tmp_path / f"shard{k}.json")) for k in range(2)]
    # initial ensemble 14 rows -> 7/7; each half-step 7 rows -> 3 on rank 0, 4 on rank 1
    assert r[0]["calls"][0] == 7 and r[1]["calls"][0] == 7
    assert set(r[0]["calls"][1:]) == {3} and set(r[1]["calls"][1:]) == {4}
    assert len(r[0]["calls"]) == 1 + 2 * steps


def test_shard_rows_cover_block():
    import bayes_skopt_amd as bask

    for B in (1, 2, 7, 128, 129):
        for ws in (1, 2, 3, 8):
            spans = [bask.distributed.shard_rows(B, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_single_process_helpers():
    import bayes_skopt_amd as bask

    d = bask.distributed
    assert d.world() == (0, 0, 1) or d.world()[2] >= 1
    c = np.arange(12.0).reshape(4, 3)
    np.testing.assert_array_equal(d.gather_chains(c), c)
    assert d.max_over_ranks(3.5) == 3.5
    assert d.rank_seed(0, 0) != d.rank_seed(0, 1)


import pytest


@pytest.mark.parametrize("tcp", ["0", "1"])
def test_unique_id_rendezvous(tmp_path, tcp):
    """The native RCCL backend's rendezvous: rank 0 hands the 128-byte ncclUniqueId to the other ranks through a file
    under /tmp (single node) or over a TCP socket on MASTER_ADDR:(MASTER_PORT + 1).  (The id itself needs a GPU:
    patched to fixed bytes here; the real ncclCommInitRank + collectives run in tests/test_gpu_distributed.py.)"""
    code = (
        "import os, sys; sys.path.insert(0, %r); import bayes_skopt_amd; from bayes_skopt_amd import _lib, distributed;"
        "_lib.comm_unique_id = lambda: bytes(range(128));"
        "r = int(os.environ['RANK']); ws = int(os.environ['WORLD_SIZE']);"
        "uid = distributed._exchange_unique_id(r, ws, timeout=60.0);"
        "open(os.path.join(%r, 'uid%%d.bin' %% r), 'wb').write(uid);"
        "import time; time.sleep(4.0 if r == 0 else 0.0)"  # (a real rank 0 stays alive in ncclCommInitRank; it removes the id file at exit)
    ) % (ROOT, str(tmp_path))
    port = _free_port()
    procs = []
    for r in (2, 0, 1):  # start order must not matter: clients retry until rank 0 listens
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="3",
                   BGP_COMM_TCP=tcp, BGP_COMM_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env))
    for p in procs:
        assert p.wait(timeout=120) == 0
    for r in range(3):
        assert open(tmp_path / f"uid{r}.bin", "rb").read() == bytes(range(128))


def test_group_falls_back_to_gloo_when_the_native_group_cannot_form(tmp_path):
    """No backend named and the native RCCL communicator cannot be created (here: patched to raise on both ranks, as
    ncclCommInitRank does collectively): the ranks say so on stderr and form a gloo group over the launcher's store;
    with BGP_DIST_BACKEND=rccl the same failure is an error."""
    code = (
        "import os, sys, json; sys.path.insert(0, %r); import numpy as np; import bayes_skopt_amd;"
        "from bayes_skopt_amd import _lib, distributed;"
        "_lib.device_count = lambda: 2; _lib.comm_available = lambda: True;\n"
        "def boom(*a, **k): raise RuntimeError('ncclCommInitRank failed: invalid usage')\n"
        "distributed._exchange_unique_id = boom\n"
        "r, lr, ws = distributed.init_process_group()\n"
        "out = distributed.gather_chains(np.full((3, 2), float(r)))\n"
        "json.dump({'backend': distributed.backend(), 'rows': out[:, 0].tolist()}, open(os.path.join(%r, 'fb%%d.json' %% r), 'w'))\n"
        "distributed.barrier(); distributed.destroy_process_group()\n"
    ) % (ROOT, str(tmp_path))
    script = tmp_path / "fallback_worker.py"
    script.write_text(code)
    base = {k: v for k, v in os.environ.items() if k != "BGP_DIST_BACKEND"}
    base.update(MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=base, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert res.stderr.count("native RCCL group failed") == 2
    for k in range(2):
        d = json.load(open(tmp_path / f"fb{k}.json"))
        assert d["backend"] == "gloo" and d["rows"] == [0.0] * 3 + [1.0] * 3
    res = subprocess.run(cmd[:-1] + [str(script)], env=dict(base, BGP_DIST_BACKEND="rccl"), capture_output=True, text=True,
                         timeout=600)
    assert res.returncode != 0 and "invalid usage" in res.stderr


def test_rendezvous_failure_on_one_rank_is_every_ranks_verdict(tmp_path):
    """A rank that cannot get the id (here: rank 1 expects another id size, so rank 0's file never satisfies it and it
    times out) reports "fail"; rank 0 -- which has its id -- must NOT walk into ncclCommInitRank alone: both raise,
    promptly, from the same status files.  The files are private (0600 in a 0700 directory of this user)."""
    code = (
        "import os, sys, json, time; sys.path.insert(0, %r); import bayes_skopt_amd; from bayes_skopt_amd import _lib, distributed;"
        "_lib.comm_unique_id = lambda: bytes(range(128));"
        "r = int(os.environ['RANK']);\n"
        "if r == 1: _lib.COMM_ID_BYTES = 64\n"
        "t0 = time.time()\n"
        "try:\n"
        "    distributed._exchange_unique_id(r, 2, timeout=3.0); out = 'entered'\n"
        "except RuntimeError as exc:\n"
        "    out = 'refused: ' + str(exc)\n"
        "modes = sorted(oct(os.stat(os.path.join(d, f)).st_mode & 0o777) for d, _s, fs in os.walk(%r) for f in fs if f.startswith('job_'))\n"
        "json.dump({'out': out, 'dt': time.time() - t0, 'modes': modes}, open(os.path.join(%r, 'v%%d.json' %% r), 'w'))\n"
    ) % (ROOT, str(tmp_path), str(tmp_path))
    script = tmp_path / "w.py"
    script.write_text(code)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r),
                                       WORLD_SIZE="2", BGP_COMM_DIR=str(tmp_path / "rdv"))) for r in (0, 1)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    v = [json.load(open(tmp_path / f"v{r}.json")) for r in range(2)]
    assert v[0]["out"].startswith("refused") and "rank 1" in v[0]["out"], v
    assert v[1]["out"].startswith("refused"), v
    assert v[0]["dt"] < 30 and v[1]["dt"] < 30
    assert set(v[0]["modes"] + v[1]["modes"]) <= {"0o600"}
    assert oct(os.stat(tmp_path / "rdv").st_mode & 0o777) == "0o700"


def test_job_names_differ_between_launches(tmp_path):
    """The id file is named after the launching process and ITS start time: a crashed job's leftover (same port, same
    world size) cannot be picked up by the next launch."""
    code = ("import os, sys; sys.path.insert(0, %r); import bayes_skopt_amd; from bayes_skopt_amd import distributed;"
            "print(distributed._job_prefix(2))") % ROOT
    env = dict(os.environ, MASTER_PORT="29500", BGP_COMM_DIR=str(tmp_path))
    env.pop("BGP_COMM_JOB", None)
    # two launches = two different parents (each `sh -c` is the common parent of its ranks)
    a = subprocess.run(["sh", "-c", f"{sys.executable} -c {code!r}; {sys.executable} -c {code!r}"], env=env,
                       capture_output=True, text=True, timeout=120).stdout.split()
    b = subprocess.run(["sh", "-c", f"{sys.executable} -c {code!r}; true"], env=env, capture_output=True, text=True,
                       timeout=120).stdout.split()
    assert len(a) == 2 and a[0] == a[1]      # ranks of one launch agree
    assert len(b) == 1 and b[0] != a[0]      # another launch: another name
    same = subprocess.run([sys.executable, "-c", code], env=dict(env, BGP_COMM_JOB="named"), capture_output=True,
                          text=True, timeout=120).stdout.split()
    assert "named" in same[0]


@pytest.mark.parametrize("n", [2, 4])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` with no launcher environment: the parent spawns N fresh rank processes (before it has
    imported the package or touched HIP), they form a group (gloo here: no GPU), rank 0's ONE line is relayed and the
    return code is the children's.  --rendezvous-only stops before the device work (which needs the MI355X)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rendezvous-only"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["dist_backend"] == "gloo" and d["ranks_seen"] == [float(r) for r in range(n)]
    assert d["max_rank"] == n - 1 and d["rccl_nranks"] is None and len(d["rank_devices"]) == n


def test_bench_self_spawn_reports_a_failing_rank():
    """A rank that dies takes the job down with its return code instead of leaving the others waiting."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    # without --rendezvous-only the ranks need a GPU: on this CPU box every rank exits with "needs an MI355X"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=600)
    if res.returncode == 0:  # (a GPU box: the job really ran)
        assert json.loads(res.stdout.splitlines()[-1])["n_gpus"] == 2
    else:
        assert "needs an MI355X" in res.stderr and not res.stdout.strip()


# ---------------------------------------------------------------------------------------------------------
# world size 8 -- the target configuration -- over gloo, and the collective's error path
# ---------------------------------------------------------------------------------------------------------
def _spawn_ranks(ws, worker, args, tmp_path, timeout=300):
    """ws plain processes with a launcher's environment (no torchrun agent: every rank's exit code is its own)."""
    port = _free_port()
    procs = []
    for r in range(ws):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(ws), LOCAL_WORLD_SIZE=str(ws), OMP_NUM_THREADS="1", BGP_DIST_BACKEND="gloo",
                   BGP_COMM_DIR=str(tmp_path / "rdv"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker)] + list(args), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs


def _toy_reference(W, steps=25):
    import bayes_skopt_amd as bask

    mu = np.array([1.0, -2.0, 0.5])

    def log_prob(Xb):
        lp = -0.5 * ((Xb - mu) ** 2).sum(axis=1)
        lp[Xb[:, 0] > 1.5] = -np.inf
        return lp

    sampler = bask.sampler.EnsembleSampler(W, 3, log_prob)
    sampler.random_state = np.random.RandomState(5).get_state()
    sampler.run_mcmc(mu + 1e-2 * np.random.RandomState(4).randn(W, 3), steps)
    return sampler.get_chain(flat=True)


@pytest.mark.parametrize("W", [256, 100])
def test_eight_rank_exact_ensemble_sharding_equals_single_process(tmp_path, W):
    """ws = 8, the north star's configuration: W = 256 walkers -> 128 proposals per half-step -> 16 rows per rank (config C's
    shard); W = 100 -> 50 rows -> an UNEVEN split (6 / 7 rows, padded to per = 7 in the exchange).  Every rank's chain is
    the single-process chain, bit for bit."""
    outs = _spawn_ranks(8, "_dist_shard_worker.py", [str(tmp_path), "toy", str(W)], tmp_path)
    for rc, _o, e in outs:
        assert rc == 0, e[-2000:]
    ref = _toy_reference(W)
    import bayes_skopt_amd as bask

    for k in range(8):
        np.testing.assert_array_equal(np.load(tmp_path / f"chain{k}.npy"), ref)
        r = json.load(open(tmp_path / f"shard{k}.json"))
        lo, hi = bask.distributed.shard_rows(W // 2, k, 8)
        assert r["ws"] == 8 and set(r["calls"][1:]) == {hi - lo} and len(r["calls"]) == 1 + 2 * 25
    if W == 256:
        assert all(json.load(open(tmp_path / f"shard{k}.json"))["calls"][1] == 16 for k in range(8))
    else:
        assert sorted(json.load(open(tmp_path / f"shard{k}.json"))["calls"][1] for k in range(8)) == [6] * 6 + [7] * 2


def test_a_rank_that_fails_inside_the_sharded_evaluation_stops_every_rank(tmp_path):
    """Rank 3 of 8 raises inside its share of a half-step.  It must still take part in the all-gather (its peers would
    block for ever otherwise -- RCCL has no time-out) and report through the status word: every rank raises
    ShardedEvaluationError naming rank 3 and exits non-zero, within seconds of the failure; rank 3's error carries the
    original exception as its cause."""
    import time

    outs = _spawn_ranks(8, "_dist_shard_worker.py", [str(tmp_path), "fail", "64"], tmp_path, timeout=240)
    t_raised = json.load(open(tmp_path / "raised.json"))["t"]
    for k, (rc, _o, e) in enumerate(outs):
        assert rc != 0, k
        assert "ShardedEvaluationError" in e, e[-1500:]
        st = json.load(open(tmp_path / f"stopped{k}.json"))
        assert "[3]" in st["msg"]
        assert st["t"] - t_raised < 30.0
        assert st["calls"] == 6  # nobody went on to another half-step
        assert ("FloatingPointError" in st["cause"]) == (k == 3)
    assert time.time() - t_raised < 120.0


def test_bench_rendezvous_at_world_size_eight():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rendezvous-only"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == [float(r) for r in range(8)] and len(d["rank_devices"]) == 8


def test_rendezvous_ignores_a_consistent_set_of_stale_files(tmp_path):
    """A killed earlier attempt under the SAME job name (a reused BGP_COMM_JOB) has left a complete, self-consistent set of
    rendezvous files behind -- id, acks, "ok" statuses of every rank.  A rank of the new attempt that starts before rank 0
    must not pick any of it up: an ack only counts with the nonce this rank has just drawn, a status only with the digest
    of the id this attempt's rank 0 created."""
    code = (
        "import os, sys, time; sys.path.insert(0, %r); import bayes_skopt_amd; from bayes_skopt_amd import _lib, distributed;"
        "_lib.comm_unique_id = lambda: bytes([7]) * 128;"
        "distributed._state['attempt'] = 1;"
        "r = int(os.environ['RANK']); time.sleep(float(os.environ['DELAY']));"
        "uid = distributed._exchange_unique_id(r, 3, timeout=60.0);"
        "open(os.path.join(%r, 'got%%d.bin' %% r), 'wb').write(uid); time.sleep(2.0 if r == 0 else 0.0)"
    ) % (ROOT, str(tmp_path))
    port = _free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="3", BGP_COMM_JOB="nightly",
                BGP_COMM_DIR=str(tmp_path / "rdv"))
    # the leftovers, written with the library's own naming (same job name, same attempt number)
    pre = ("import os, sys; sys.path.insert(0, %r); import bayes_skopt_amd; from bayes_skopt_amd import distributed;"
           "distributed._state['attempt'] = 1; p = distributed._job_prefix(3); old = bytes([9]) * 128;"
           "dg = distributed._uid_digest(old);"
           "[distributed._write_private(p + '.ack.%%d' %% r, b'0' * 32 + old) for r in (1, 2)];"
           "[distributed._write_private(p + '.hello.%%d' %% r, b'0' * 32 + b'|0000:01:00.0') for r in (1, 2)];"
           "[distributed._write_private(p + '.st.%%d' %% r, b'ok:' + dg) for r in (0, 1, 2)];"
           "distributed._write_private(p + '.uid', old)") % ROOT
    assert subprocess.run([sys.executable, "-c", pre], env=base, timeout=120).returncode == 0
    old = time_old = None
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(base, RANK=str(r), DELAY=("3.0" if r == 0 else "0.0")))
             for r in (1, 2, 0)]  # ranks 1 and 2 are up (and polling) three seconds before rank 0 cleans up
    for p in procs:
        assert p.wait(timeout=180) == 0
    for r in range(3):
        assert open(tmp_path / f"got{r}.bin", "rb").read() == bytes([7]) * 128


# ---------------------------------------------------------------------------------------------------------
# which backend, which device: an 8-GPU node as the driver launches it, and as a launcher that pins one GPU per rank does
# ---------------------------------------------------------------------------------------------------------
def test_plan_group_for_eight_ranks_on_an_eight_device_node():
    """``distributed.plan_group``: pure host logic, decided alike on every rank from what it sees (no exchange)."""
    import bayes_skopt_amd  # noqa: F401
    from bayes_skopt_amd import distributed as D

    # torch.distributed.run --nproc-per-node 8 on an 8-GPU node: every rank sees 8 devices, rank r takes GPU r
    assert [D.plan_group(8, lr, 8, True, pinned=False) for lr in range(8)] == [("rccl", lr) for lr in range(8)]
    # ... and 4 ranks on the same node take GPUs 0..3
    assert [D.plan_group(8, lr, 4, True, pinned=False) for lr in range(4)] == [("rccl", lr) for lr in range(4)]
    # a launcher that pins one GPU per rank (HIP_VISIBLE_DEVICES=<local rank>): every rank sees ONE device, ordinal 0 -- the
    # native group is tried and the rendezvous tells from the PCI bus ids whether those are eight GPUs (below)
    assert [D.plan_group(1, lr, 8, True, pinned=True) for lr in range(8)] == [("rccl", 0)] * 8
    # two visible devices per rank for eight ranks, pinned: local_rank % 2 (the rendezvous will find the shared ones)
    assert [D.plan_group(2, lr, 8, True, pinned=True)[1] for lr in range(8)] == [0, 1] * 4
    # ranks that share the one GPU of a box and no pinning (the GPU tests of this repository): gloo, no attempt
    assert D.plan_group(1, 1, 2, True, pinned=False) == ("gloo", None)
    # no device (the CPU tests), or no librccl: gloo
    assert D.plan_group(0, 0, 2, False, pinned=False) == ("gloo", None) and D.plan_group(8, 3, 8, False, pinned=False) == ("gloo", None)
    # by name: taken as asked; rccl without a device is an error (no CPU fallback), an unknown name too
    assert D.plan_group(8, 5, 8, True, name="gloo") == ("gloo", None)
    assert D.plan_group(1, 1, 2, True, name="rccl", pinned=False) == ("rccl", 0)
    assert D.plan_group(8, 5, 8, True, name="rccl", device=2) == ("rccl", 2)
    with pytest.raises(RuntimeError):
        D.plan_group(0, 0, 2, False, name="rccl")
    with pytest.raises(ValueError):
        D.plan_group(8, 0, 8, True, name="nccl")
    # the environment decides `pinned` when it is not given
    old = {v: os.environ.pop(v, None) for v in D._VISIBLE_DEVICES_VARS}
    try:
        assert D.plan_group(1, 3, 8, True) == ("gloo", None)
        os.environ["HIP_VISIBLE_DEVICES"] = "3"
        assert D.plan_group(1, 3, 8, True) == ("rccl", 0)
    finally:
        os.environ.pop("HIP_VISIBLE_DEVICES", None)
        os.environ.update({k: v for k, v in old.items() if v is not None})


@pytest.mark.parametrize("shared", [False, True])
def test_rendezvous_tells_eight_pinned_gpus_from_a_shared_one(tmp_path, shared):
    """Eight ranks that all call their GPU "device 0" (a launcher pinned one per rank): the status files carry the devices' PCI
    bus ids; eight different ones -> every rank gets the id; two ranks on the same one -> EVERY rank refuses, before anybody
    has entered ncclCommInitRank, naming the two."""
    code = (
        "import os, sys, json; sys.path.insert(0, %r); import bayes_skopt_amd; from bayes_skopt_amd import _lib, distributed;"
        "_lib.comm_unique_id = lambda: bytes(range(128));"
        "r = int(os.environ['RANK']);\n"
        "bus = '0000:%%02x:00.0' %% (5 if (%r and r == 6) else r)\n"
        "try:\n"
        "    uid = distributed._exchange_unique_id(r, 8, timeout=60.0, device_id=bus); out = 'id ' + uid.hex()[:8]\n"
        "except RuntimeError as exc:\n"
        "    out = 'refused: ' + str(exc)\n"
        "json.dump({'out': out}, open(os.path.join(%r, 'v%%d.json' %% r), 'w'))\n"
        "import time; time.sleep(4.0 if r == 0 else 0.0)\n"  # (a real rank 0 stays alive in ncclCommInitRank; it removes the files at exit)
    ) % (ROOT, shared, str(tmp_path))
    script = tmp_path / "w.py"
    script.write_text(code)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="8",
                                       BGP_COMM_DIR=str(tmp_path / "rdv"), BGP_COMM_JOB="pin%d" % shared)) for r in range(8)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    outs = [json.load(open(tmp_path / f"v{r}.json"))["out"] for r in range(8)]
    if shared:
        assert all(o.startswith("refused") and "ranks 5 and 6 share device 0000:05:00.0" in o for o in outs), outs
    else:
        assert outs == ["id 00010203"] * 8, outs


def _stub_ranks(ws, mode, args, tmp_path, timeout=600):
    """`ws` rank processes of tests/_stub_rccl_bench.py: the native ("rccl") host path with a file-backed stand-in communicator
    and a closed-form stand-in context (no GPU)."""
    xdir = tmp_path / "x"
    xdir.mkdir()
    port = _free_port()
    procs = []
    for r in range(ws):
        env = {k: v for k, v in os.environ.items() if k not in ("BGP_DIST_BACKEND", "BGP_DIST_FORCE")}
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ws),
                   LOCAL_WORLD_SIZE=str(ws), OMP_NUM_THREADS="1", BGP_COMM_DIR=str(tmp_path / "rdv"), BGP_COMM_JOB="stub")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_stub_rccl_bench.py"), str(xdir), mode] + list(args),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs, xdir


@pytest.mark.parametrize("shard", ["ensemble", "chains"])
def test_bench_two_ranks_through_the_native_backends_host_side(tmp_path, shard):
    """`bench.py --gpus 2 --shard ensemble | chains` through the RCCL code path's HOST side with world = 2 semantics (a stand-in
    communicator and context, tests/_stub_rccl_bench.py): the group that forms is reported as the native one (`dist_backend`,
    `rccl_nranks` = 2, a device per rank out of the 8 visible ones), the sharded log-probability takes its native branch (submit ->
    collective with status words), the line carries the N > 1 keys."""
    outs, _ = _stub_ranks(2, "bench", ["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--shard", shard], tmp_path)
    assert [rc for rc, _o, _e in outs] == [0, 0], outs[0][2][-1500:] + outs[1][2][-1500:]
    lines = [ln for ln in outs[0][1].splitlines() if ln.strip()]
    assert len(lines) == 1 and not outs[1][1].strip()
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist_backend"] == "rccl" and d["rccl_nranks"] == 2 and d["rank_devices"] == [0, 1]
    assert d["resident"] is False and d["sampler"].startswith("host-driven") and len(d["timed_passes_ms_per_step"]) == 3
    assert d["collective_ms_per_half_step"] == 0.005 and d["collective_note"].startswith("in-stream")
    if shard == "ensemble":
        assert d["scaling"] == "strong" and d["config"]["proposals_per_gpu_per_half_step"] == 64 and d["weak_chains_evals_per_s"] > 0
        assert d["config"]["walkers_total"] == 256 and d["gathered_chain_rows"] == 2 * 256
    else:
        assert d["scaling"] == "weak" and d["strong_ensemble_evals_per_s"] > 0 and d["gathered_chain_rows"] == 2 * 2 * 256


def test_a_rank_interrupted_inside_a_half_step_takes_the_native_group_down(tmp_path):
    """KeyboardInterrupt on rank 1 between submit and collective: it collects its pending batch, ABORTS the communicator
    (`abort_process_group` -> `Comm.abort`, bgp_comm_abort on the real one) and goes down; rank 0, waiting in the collective,
    fails at once with the communicator's error instead of sitting there for BGP_COMM_TIMEOUT_S."""
    outs, xdir = _stub_ranks(2, "interrupt", [], tmp_path)
    r = [json.load(open(xdir / f"r{k}.json")) for k in range(2)]
    assert [rc for rc, _o, _e in outs] == [3, 3]
    assert r[1]["out"] == "interrupted" and r[1]["aborted"]
    assert r[0]["out"].startswith("BgpError") and "communicator aborted" in r[0]["out"] and r[0]["dt"] < 30
