"""Multi-GPU sharding of the MCMC chains: one process per GPU (launched by ``torch.distributed.run`` or any other
launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).

Default (SURVEY.md 8(e) option 2 / BASELINE.json north_star: "chains shard naturally ... RCCL over xGMI
only for the final posterior-sample gather"): each rank runs an independent sub-ensemble on its own
device with NO collective in the sampling loop; the only exchange is the final gather of the posterior
samples (``gather_chains``).

Option (SURVEY.md 8(e) option 1, exact single-ensemble semantics): every rank holds the same data and
the same sampler RNG, so all ranks propose the same (B, p) block each half-step; ``shard_log_prob``
makes rank r evaluate rows [r*B/G, (r+1)*B/G) on its device and all-gathers the B log-probabilities
(one latency-bound collective of B doubles per half-step).  Accept/reject then runs identically on
every rank and the chain equals the single-GPU chain bit for bit.

Backends (``BGP_DIST_BACKEND`` or the ``backend`` argument):
  ``rccl``  (default when this process sees a GPU) -- RCCL through libbgp's own C-ABI (``bgp_comm_*``,
            csrc/bgp_comm.hip): no PyTorch anywhere in the product path.  Rank 0 creates the ncclUniqueId and
            hands its 128 bytes to the other ranks through a file under /tmp on a single node (MASTER_ADDR =
            loopback) or over a TCP socket on MASTER_ADDR:(MASTER_PORT + 1) otherwise (``BGP_COMM_TCP=1`` forces it,
            ``BGP_COMM_PORT`` overrides the port; MASTER_PORT itself belongs to the launcher's rendezvous store).
  ``gloo`` / ``nccl`` -- ``torch.distributed`` process groups: the CPU tests (world size 2 over gloo) and an A/B
            path for the native one; torch is only imported when one of these is selected.
When no backend is named and the native group cannot be formed (librccl missing, id exchange timed out,
ncclCommInitRank refused the group) every rank reports it on stderr and the job continues over gloo: the only exchange
on the default path is the final gather of host-resident chains.  A backend asked for by name fails loudly instead.
``BGP_DIST_FORCE=1`` joins a group even at world size 1 (the GPU test that runs real RCCL collectives on one GPU).
"""
import os
import socket
import time

import numpy as np

__all__ = ["world", "init_process_group", "destroy_process_group", "gather_chains", "barrier", "max_over_ranks",
           "rank_seed", "shard_rows", "shard_log_prob", "broadcast_array", "backend"]

_state = {"backend": None, "comm": None, "rank": 0, "world": 1}


def world():
    """(rank, local_rank, world_size) from the launcher's environment (1 process when absent)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def backend():
    """Name of the active backend ("rccl", "gloo", "nccl") or None outside a group."""
    return _state["backend"]


class _stdout_to_stderr:
    """Native libraries announce themselves on file descriptor 1 ("[Gloo] Rank 0 is connected to ...", RCCL's version
    banner); a job's standard output belongs to its caller (bench.py: one JSON line).  While a group is being formed,
    descriptor 1 points at standard error."""

    def __enter__(self):
        import sys

        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import sys

        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _torch_dist():
    import torch.distributed as dist

    return dist


# ---------------------------------------------------------------------------------------------------------
# rendezvous of the ncclUniqueId for the native backend
# ---------------------------------------------------------------------------------------------------------
def _comm_endpoint():
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("BGP_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
    return host, port


_T_START = time.time()


def _uid_file():
    tag = "%s_%s_%s_ws%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "29500"),
                             os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("WORLD_SIZE", "1"))
    return os.path.join(os.environ.get("BGP_COMM_DIR", "/tmp"), "bgp_comm_uid_" + tag.replace("/", "_"))


def _exchange_unique_id(rank, ws, timeout=300.0):
    """Rank 0's ncclUniqueId on every rank.  Single node (MASTER_ADDR is this host's loopback, what the launcher
    contract uses): rank 0 drops the bytes into /tmp/bgp_comm_uid_<addr>_<port>_<run id> (atomic rename) and the
    other ranks poll for a file no older than two minutes before their own start (rank 0 removes it at exit; a stale
    one left by a crashed job with the same port and world size is ignored); no port beyond the launcher's own is needed.  Otherwise: a TCP socket on MASTER_ADDR:(MASTER_PORT+1)."""
    from . import _lib

    local = os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1") \
        and os.environ.get("BGP_COMM_TCP") != "1"
    if rank == 0:
        uid = _lib.comm_unique_id()
        if ws == 1:
            return uid
        if local:
            path = _uid_file()
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "wb") as f:
                f.write(uid)
            os.replace(tmp, path)
            import atexit

            atexit.register(lambda p=path: os.path.exists(p) and os.remove(p))  # no stale id for the next job
            return uid
        host, port = _comm_endpoint()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((host, port))
        srv.listen(ws)
        srv.settimeout(timeout)
        try:
            for _ in range(ws - 1):
                conn, _addr = srv.accept()
                with conn:
                    conn.sendall(uid)
        finally:
            srv.close()
        return uid
    deadline = time.monotonic() + timeout
    if local:
        path = _uid_file()
        while True:
            try:
                if os.path.getmtime(path) >= _T_START - 120.0:  # (ranks of one job start within seconds of each other)
                    with open(path, "rb") as f:
                        buf = f.read()
                    if len(buf) == _lib.COMM_ID_BYTES:
                        return buf
            except OSError:
                pass
            if time.monotonic() > deadline:
                raise RuntimeError(f"rank {rank}: no ncclUniqueId file {path} from rank 0 within {timeout:.0f} s")
            time.sleep(0.02)
    host, port = _comm_endpoint()
    while True:
        try:
            with socket.create_connection((host, port), timeout=5.0) as s:
                buf = b""
                while len(buf) < _lib.COMM_ID_BYTES:
                    part = s.recv(_lib.COMM_ID_BYTES - len(buf))
                    if not part:
                        raise ConnectionError("rank 0 closed the id socket early")
                    buf += part
                return buf
        except (ConnectionRefusedError, ConnectionError, socket.timeout, OSError):
            if time.monotonic() > deadline:
                raise RuntimeError(f"rank {rank}: no ncclUniqueId from rank 0 at {host}:{port} within {timeout:.0f} s")
            time.sleep(0.05)


def init_process_group(backend=None, device=None):
    """Join the process group when launched with WORLD_SIZE > 1 (or BGP_DIST_FORCE=1).  Returns
    (rank, local_rank, world_size)."""
    rank, local_rank, ws = world()
    if _state["backend"] is not None or (ws <= 1 and os.environ.get("BGP_DIST_FORCE") != "1"):
        return rank, local_rank, ws
    from . import _lib

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = _lib.device_count()
    name = backend or os.environ.get("BGP_DIST_BACKEND")
    if not name:
        # librccl loads (or not) identically on every rank of a node: a consistent choice without any exchange
        name = ("rccl" if _lib.comm_available() else "nccl") if ndev > 0 else "gloo"
    explicit = bool(backend or os.environ.get("BGP_DIST_BACKEND"))
    if name == "rccl":
        if ndev < 1:
            raise RuntimeError("BGP_DIST_BACKEND=rccl needs an MI355X (no CPU fallback); use gloo for CPU tests")
        dev = (local_rank % ndev) if device is None else int(device)
        try:
            uid = _exchange_unique_id(rank, ws)
            _state["comm"] = _lib.Comm(dev, rank, ws, uid)
        except Exception as exc:
            # ncclCommInitRank is collective: it fails on every rank or on none.  When the backend was not asked for
            # by name, the job goes on with the launcher's own store and gloo (the only exchange is the final gather of
            # the chains: host memory either way); an explicit BGP_DIST_BACKEND=rccl fails loudly.
            if explicit:
                raise
            import sys

            print(f"[bayes_skopt_amd.distributed] rank {rank}: native RCCL group failed ({exc}); using gloo",
                  file=sys.stderr, flush=True)
            name = "gloo"
    if name != "rccl":
        import datetime

        import torch

        dist = _torch_dist()
        if not dist.is_initialized():
            if name == "nccl":
                torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            with _stdout_to_stderr():
                dist.init_process_group(backend=name, rank=rank, world_size=ws, timeout=datetime.timedelta(seconds=300))
                dist.barrier()  # (gloo connects its pairs lazily: make it talk now, while descriptor 1 is redirected)
    _state.update(backend=name, rank=rank, world=ws)
    return rank, local_rank, ws


def destroy_process_group():
    if _state["backend"] == "rccl":
        _state["comm"].close()
    elif _state["backend"] is not None:
        dist = _torch_dist()
        if dist.is_initialized():
            dist.destroy_process_group()
    _state.update(backend=None, comm=None, rank=0, world=1)


def _torch_device():
    import torch

    if _state["backend"] == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _allgather(a):
    """(world,) + a.shape, rank-major, on every rank."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if _state["backend"] == "rccl":
        return _state["comm"].allgather(a)
    import torch

    dist = _torch_dist()
    t = torch.from_numpy(a.copy()).to(_torch_device())
    out = [torch.empty_like(t) for _ in range(_state["world"])]
    dist.all_gather(out, t)
    return torch.stack(out, dim=0).cpu().numpy()


def rank_seed(seed, rank):
    """Distinct, reproducible RNG seed per rank (independent sub-ensembles)."""
    return int((int(seed) * 1000003 + 7919 * int(rank)) % (2**31 - 1))


def gather_chains(chain):
    """All-gather equally shaped per-rank chains (S, p) -> (world*S, p), rank-major, on every rank."""
    chain = np.ascontiguousarray(chain, dtype=np.float64)
    if _state["backend"] is None:
        return chain
    return _allgather(chain).reshape((-1,) + chain.shape[1:])


def barrier():
    if _state["backend"] == "rccl":
        _state["comm"].barrier()
    elif _state["backend"] is not None:
        _torch_dist().barrier()


def max_over_ranks(value):
    """MAX all-reduce of a python float (timing)."""
    if _state["backend"] is None:
        return float(value)
    if _state["backend"] == "rccl":
        return float(_state["comm"].allreduce_max([float(value)])[0])
    import torch

    dist = _torch_dist()
    t = torch.tensor([float(value)], dtype=torch.float64, device=_torch_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_rows(B, rank, ws):
    """Row range [lo, hi) of a B-row proposal block owned by `rank` (contiguous, sizes differ by <= 1)."""
    return (B * rank) // ws, (B * (rank + 1)) // ws


def shard_log_prob(fn):
    """Wrap a vectorised log-probability ``fn(Theta (B,p), **kw) -> (B,)`` so that each rank evaluates only
    its own rows and the full vector is re-assembled with one all-gather (exact single-ensemble sharding).
    Every rank must call it with the same Theta (same sampler RNG on every rank).  Outside a process
    group it is ``fn`` itself."""

    def wrapped(Theta, *args, **kwargs):
        Theta = np.atleast_2d(np.asarray(Theta, dtype=np.float64))
        if _state["backend"] is None:
            return fn(Theta, *args, **kwargs)
        ws, rank = _state["world"], _state["rank"]
        B = Theta.shape[0]
        lo, hi = shard_rows(B, rank, ws)
        local = np.zeros(-(-B // ws))
        if hi > lo:
            local[: hi - lo] = fn(Theta[lo:hi], *args, **kwargs)
        parts = _allgather(local)
        full = np.empty(B)
        for r in range(ws):
            rlo, rhi = shard_rows(B, r, ws)
            full[rlo:rhi] = parts[r][: rhi - rlo]
        return full

    return wrapped


def broadcast_array(arr, src=0):
    """Broadcast a float64 array from rank `src` (identity outside a process group).  Used to pin the
    start ensemble of the sharded sampler to one rank's copy."""
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    if _state["backend"] is None:
        return arr
    if _state["backend"] == "rccl":
        return _state["comm"].broadcast(arr, root=src).reshape(arr.shape)
    import torch

    t = torch.from_numpy(arr.copy()).to(_torch_device())
    _torch_dist().broadcast(t, src=src)
    return t.cpu().numpy()
