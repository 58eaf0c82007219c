"""Test infrastructure, NOT the product: `bench.py` (or a sharded fit) as a rank of a world-size-N job whose NATIVE backend
("rccl": bayes-skopt_amd/distributed.py's own branches, `_lib.Comm`, `Context.lml_wait_allgather`) is served by stand-ins that
need no GPU -- a communicator that exchanges through files and a context whose "log-likelihood" is a cheap closed form.  What
runs for real is the HOST side of the multi-GPU path with world > 1 semantics: backend / device planning (8 visible devices),
the id rendezvous with device identities, `group_info` (rccl_nranks, rank_devices), the sharded log-probability's native branch
(submit -> collective with a status word -> every rank raises alike), `abort_process_group`, and bench.py's N > 1 line.

    python tests/_stub_rccl_bench.py <exchange dir> bench [bench.py arguments ...]
    python tests/_stub_rccl_bench.py <exchange dir> interrupt      # rank 1 is interrupted inside a half-step
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bayes_skopt_amd  # noqa: E402,F401
from bayes_skopt_amd import _lib, bayesgpr, distributed  # noqa: E402

XDIR = sys.argv[1]
NDEV = 8


class StubComm:
    """`_lib.Comm` over files in XDIR: every collective is a numbered round, every rank writes its part and reads all."""

    live = []

    def __init__(self, device, rank, world, unique_id):
        assert bytes(unique_id) == bytes(range(128)) and 0 <= device < NDEV
        self.device, self.rank, self.world = int(device), int(rank), int(world)
        self._seq = 0
        self._h = object()
        self.aborted = False
        StubComm.live.append(self)

    def _exchange(self, a):
        if self.aborted:
            raise _lib.BgpError("collective failed (code 6): the communicator was aborted by an earlier failed collective")
        a = np.ascontiguousarray(a, dtype=np.float64)
        self._seq += 1
        tmp = os.path.join(XDIR, "c%d.%d.tmp.npy" % (self._seq, self.rank))
        np.save(tmp, a)
        os.replace(tmp, os.path.join(XDIR, "c%d.%d.npy" % (self._seq, self.rank)))
        parts, t0 = [], time.monotonic()
        for r in range(self.world):
            path = os.path.join(XDIR, "c%d.%d.npy" % (self._seq, r))
            while not os.path.exists(path):
                if os.path.exists(os.path.join(XDIR, "ABORT")) or time.monotonic() - t0 > 60:
                    self.aborted = True  # (ncclCommAbort by a peer, or BGP_COMM_TIMEOUT_S: BGP_ERR_COMM)
                    raise _lib.BgpError("collective failed (code 6): a peer has died or left; communicator aborted")
                time.sleep(0.001)
            parts.append(np.load(path))
        return np.stack(parts)

    def allgather(self, a):
        return self._exchange(a)

    def allreduce_max(self, a):
        return self._exchange(np.array(a, dtype=np.float64)).max(axis=0)

    def broadcast(self, a, root=0):
        return self._exchange(np.array(a, dtype=np.float64))[root]

    def barrier(self):
        self._exchange(np.zeros(1))

    def nranks(self):
        return self.world

    def abort(self):
        self.aborted = True
        open(os.path.join(XDIR, "ABORT"), "w").close()

    def bench_lml_gather(self, ctx, per, reps=200):
        self._exchange(np.zeros(per + 1))
        return 0.005

    def close(self):
        pass


def fake_lml(H):
    """A smooth, finite stand-in for the log-likelihood of canonical vectors (the host plumbing is what is under test)."""
    H = np.atleast_2d(H)
    return -0.5 * np.sum((H - np.linspace(-1.0, 0.5, H.shape[1])) ** 2, axis=1) - 3.0


class StubContext:
    def __init__(self, X, y, alpha_diag, form="product", stationary="matern52", max_batch=64, device=0):
        self.n, self.d = np.atleast_2d(X).shape
        self.p, self.form, self.stationary = self.d + 2, form, stationary
        self.max_batch, self.device = int(max_batch), int(device)
        self._pending, self._timing, self.resident_H = None, False, None
        self._h = object()

    def close(self):
        pass

    def update_data(self, X, y, alpha_diag):
        self.n = np.atleast_2d(X).shape[0]

    def set_warp(self, w):
        pass

    def set_streams(self, n):
        pass

    def set_persist(self, mode):
        pass

    def set_timing(self, enable):
        self._timing = bool(enable)

    def lml(self, H, return_status=False):
        out = fake_lml(H)
        return (out, np.zeros(len(out), dtype=np.int32)) if return_status else out

    def lml_submit(self, H):
        H = np.atleast_2d(H)
        if len(H) > self.max_batch or len(H) == 0 or self._timing:
            return False
        self._pending = H.copy()
        return True

    def has_pending(self):
        return self._pending is not None

    def lml_wait(self, return_status=False):
        H, self._pending = self._pending, None
        return self.lml(H, return_status)

    def lml_wait_allgather(self, comm, per_rank, local_error=0):
        """bgp_lml_batch_wait_allgather: per_rank values + ONE status word per rank; the pending batch is consumed whatever
        happens; a local failure travels as its code beside NaN values."""
        H, self._pending = self._pending, None
        slot = np.full(per_rank + 1, np.nan)
        if not local_error and H is not None:
            slot[: len(H)] = fake_lml(H)
        slot[per_rank] = float(local_error)
        g = comm.allgather(slot)
        return g[:, :per_rank], g[:, per_rank].astype(np.int32)

    def posterior(self, H, want_L=False, want_alpha=True, want_K_inv=False):
        B = len(np.atleast_2d(H))
        return {"L": None, "alpha": np.zeros((B, self.n)), "K_inv": None, "lml": fake_lml(H), "status": np.zeros(B, dtype=np.int32)}

    def last_timing(self):
        t = {k: {"ms": 1.0, "launches": 1} for k in ("kbuild", "potrf", "trsm", "syrk", "syrk_columns")}
        t["device_total_ms"] = 4.0
        return t

    def gen_stats(self):
        return {"batches": 0, "launches": 0}

    def persist_stats(self):
        return {"calls": 0, "timeouts": 0, "disabled": False, "cooldown_left": 0}


_lib.device_count = lambda: NDEV
_lib.comm_available = lambda: True
_lib.comm_unique_id = lambda: bytes(range(128))
_lib.device_identity = lambda dev=0: "0000:%02x:00.0" % (16 + int(dev))
_lib.device_synchronize = lambda device=0: None
_lib.Comm = StubComm
_lib.Context = StubContext
bayesgpr._resident_run = lambda *a, **k: (None, None)  # (no device: the host-driven loop, silently)

mode = sys.argv[2]
if mode == "bench":
    import bench

    sys.argv = ["bench.py"] + sys.argv[3:]
    bench.main()
elif mode == "interrupt":
    import json

    import bayes_skopt_amd as bask

    rank, _lr, ws = distributed.init_process_group()
    assert distributed.backend() == "rccl"
    rng = np.random.RandomState(0)
    X = rng.uniform(size=(40, 2))
    y = np.sin(3.0 * X.sum(axis=1))
    calls = [0]

    def prior(t):
        calls[0] += 1
        if rank == 1 and calls[0] > 30:
            raise KeyboardInterrupt  # this rank goes down OUTSIDE a collective: it must take the group with it
        return -0.5 * np.asarray(t) ** 2

    gp = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1]), random_state=3, shard_ensemble=True, optimizer=None)
    t0 = time.time()
    try:
        gp.fit(X, y, n_desired_samples=20 * 30, n_burnin=0, n_walkers_per_thread=20, progress=False, priors=[prior] * 4)
        out = "finished"
    except KeyboardInterrupt:
        out = "interrupted"
    except Exception as exc:  # noqa: BLE001
        out = "%s: %s" % (type(exc).__name__, exc)
    json.dump({"out": out, "dt": time.time() - t0, "aborted": bool(StubComm.live and StubComm.live[0].aborted)},
              open(os.path.join(XDIR, "r%d.json" % rank), "w"))
    sys.exit(0 if out == "finished" else 3)
