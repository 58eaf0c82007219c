"""Multi-GPU sharding of the MCMC chains: one process per GPU (``torch.distributed.run``).

Default (SURVEY.md 8(e) option 2 / BASELINE.json north_star: "chains shard naturally ... RCCL over xGMI
only for the final posterior-sample gather"): each rank runs an independent sub-ensemble on its own
device with NO collective in the sampling loop; the only exchange is the final gather of the posterior
samples (``gather_chains``: RCCL all-gather over xGMI on GPUs, gloo in the CPU tests).

Option (SURVEY.md 8(e) option 1, exact single-ensemble semantics): every rank holds the same data and
the same sampler RNG, so all ranks propose the same (B, p) block each half-step; ``shard_log_prob``
makes rank r evaluate rows [r*B/G, (r+1)*B/G) on its device and all-gathers the B log-probabilities
(one latency-bound collective of B doubles per half-step).  Accept/reject then runs identically on
every rank and the chain equals the single-GPU chain bit for bit.

torch is used here for process-group plumbing only (rendezvous, RCCL); nothing in the numerical path
touches it.
"""
import os

import numpy as np

__all__ = ["world", "init_process_group", "gather_chains", "barrier", "max_over_ranks", "rank_seed", "shard_rows",
           "shard_log_prob", "broadcast_array"]


def world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process when absent)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def _dist():
    import torch.distributed as dist

    return dist


def init_process_group(backend=None):
    """Join the process group when launched with WORLD_SIZE > 1 (backend "nccl" == RCCL on ROCm;
    "gloo" for CPU tests).  Returns (rank, local_rank, world_size)."""
    rank, local_rank, ws = world()
    if ws > 1:
        import torch

        dist = _dist()
        if not dist.is_initialized():
            if backend is None:
                backend = os.environ.get("BGP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend == "nccl":
                torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return rank, local_rank, ws


def _is_dist():
    try:
        dist = _dist()
    except Exception:
        return False
    return dist.is_available() and dist.is_initialized()


def _device_for_backend():
    import torch

    dist = _dist()
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def rank_seed(seed, rank):
    """Distinct, reproducible RNG seed per rank (independent sub-ensembles)."""
    return int((int(seed) * 1000003 + 7919 * int(rank)) % (2**31 - 1))


def gather_chains(chain):
    """All-gather equally shaped per-rank chains (S, p) -> (world*S, p), rank-major, on every rank."""
    chain = np.ascontiguousarray(chain, dtype=np.float64)
    if not _is_dist():
        return chain
    import torch

    dist = _dist()
    ws = dist.get_world_size()
    dev = _device_for_backend()
    t = torch.from_numpy(chain).to(dev)
    out = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(out, t)
    return torch.cat(out, dim=0).cpu().numpy()


def barrier():
    if _is_dist():
        _dist().barrier()


def max_over_ranks(value):
    """MAX all-reduce of a python float (timing)."""
    if not _is_dist():
        return float(value)
    import torch

    dist = _dist()
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device_for_backend())
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
        if not _is_dist():
            return fn(Theta, *args, **kwargs)
        import torch

        dist = _dist()
        ws, rank = dist.get_world_size(), dist.get_rank()
        B = Theta.shape[0]
        lo, hi = shard_rows(B, rank, ws)
        chunk = -(-B // ws)
        local = np.zeros(chunk)
        if hi > lo:
            local[: hi - lo] = fn(Theta[lo:hi], *args, **kwargs)
        dev = _device_for_backend()
        t = torch.from_numpy(local).to(dev)
        out = [torch.empty_like(t) for _ in range(ws)]
        dist.all_gather(out, t)
        full = np.empty(B)
        for r in range(ws):
            rlo, rhi = shard_rows(B, r, ws)
            if rhi > rlo:
                full[rlo:rhi] = out[r][: rhi - rlo].cpu().numpy()
        return full

    return wrapped


def broadcast_array(arr, src=0):
    """Broadcast a float64 array from rank `src` (identity outside a process group).  Used to pin the
    start ensemble of the sharded sampler to one rank's copy."""
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    if not _is_dist():
        return arr
    import torch

    t = torch.from_numpy(arr.copy()).to(_device_for_backend())
    _dist().broadcast(t, src=src)
    return t.cpu().numpy()
