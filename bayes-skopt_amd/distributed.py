"""Multi-GPU sharding of the hyper-posterior MCMC: one process per GPU (launched by ``torch.distributed.run``, by
``bench.py --gpus N`` itself, or by any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).

Exact single-ensemble sharding (SURVEY.md 8(e) option 1; what ``bask/bayesgpr.py:490-530`` runs is ONE
``n_walkers`` ensemble on one RNG): every rank holds the same data and drives the same sampler RNG, so all ranks
propose the same (B, p) block each half-step; rank r evaluates rows [r*B/G, (r+1)*B/G) on its device and the B
log-likelihoods are all-gathered (``ShardedLogProb`` in bayesgpr.py through ``allgather_lml``: device to device over RCCL,
one latency-bound collective of B doubles per half-step).  Accept/reject then runs identically on every rank and the
chain equals the single-GPU chain bit for bit.

Independent sub-ensembles (SURVEY.md 8(e) option 2): each rank its own walkers and seeds, NO collective in the
sampling loop; the only exchange is the final gather of the posterior samples (``gather_chains``).

Backends (``BGP_DIST_BACKEND`` or the ``backend`` argument):
  ``rccl``  (default when every rank has a GPU of its own) -- RCCL through libbgp's own C-ABI (``bgp_comm_*``,
            csrc/bgp_comm.hip): no PyTorch anywhere in the product path.  Rank 0 creates the ncclUniqueId and hands its
            128 bytes to the other ranks through a file that is unique to the job (named after the common launcher
            process and its start time, ``BGP_COMM_JOB`` overrides) in a per-user 0700 directory; every rank then
            publishes "have it" / "failed" and nobody enters ncclCommInitRank before all ranks have it -- one rank
            timing out cannot leave the others blocked in the collective.  Across nodes (MASTER_ADDR not loopback, or
            ``BGP_COMM_TCP=1``): a TCP socket on MASTER_ADDR:(MASTER_PORT + 1) (``BGP_COMM_PORT`` overrides the port;
            MASTER_PORT itself belongs to the launcher's rendezvous store).
  ``gloo``  -- a ``torch.distributed`` CPU group: the world-size-2 CPU tests, and ranks that share one GPU (RCCL refuses
            two ranks on a device); torch is only imported when this is selected.  There is no second GPU backend.
When no backend is named and the native group cannot be formed (librccl missing, id exchange failed on ANY rank) every
rank learns so from the same status files, says so on stderr and the job continues over gloo.  A backend asked for by
name fails loudly instead.  ``BGP_DIST_FORCE=1`` joins a group even at world size 1 (the GPU test that runs real RCCL
collectives on one GPU).
"""
import os
import socket
import sys
import tempfile
import time

import numpy as np

__all__ = ["world", "plan_group", "init_process_group", "destroy_process_group", "gather_chains", "barrier", "max_over_ranks",
           "rank_seed", "shard_rows", "shard_log_prob", "broadcast_array", "backend", "group_info", "allgather_lml",
           "ShardedEvaluationError"]

_state = {"backend": None, "comm": None, "rank": 0, "world": 1, "device": None, "job_prefix": None, "attempt": 0}


class ShardedEvaluationError(RuntimeError):
    """A rank's share of a sharded proposal block failed.  Raised on EVERY rank of the group from the same gathered status
    words (the failing rank still took part in the collective), so that the whole job stops instead of hanging."""


def world():
    """(rank, local_rank, world_size) from the launcher's environment (1 process when absent)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def backend():
    """Name of the active backend ("rccl", "gloo") or None outside a group."""
    return _state["backend"]


def communicator():
    """The native group's ``_lib.Comm`` (None outside an RCCL group): what the sharded resident sampler enqueues its per-half-step
    all-gather on."""
    return _state["comm"] if _state["backend"] == "rccl" else None


class _stdout_to_stderr:
    """Native libraries announce themselves on file descriptor 1 ("[Gloo] Rank 0 is connected to ...", RCCL's version
    banner); a job's standard output belongs to its caller (bench.py: one JSON line).  While a group is being formed,
    descriptor 1 points at standard error."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
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


def _job_id():
    """A name every rank of THIS launch derives alike and no other launch shares: the common parent (torchrun's agent,
    bench.py's self-spawning parent, a shell) and that process's start time in clock ticks (/proc/<pid>/stat field
    22) -- a recycled pid has another start time, a crashed earlier job another parent.  ``BGP_COMM_JOB`` names it
    explicitly for launchers whose ranks have no common parent."""
    job = os.environ.get("BGP_COMM_JOB")
    if job:
        return job
    ppid = os.getppid()
    try:
        with open("/proc/%d/stat" % ppid) as f:
            start = f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        start = "0"
    return "p%d_%s" % (ppid, start)


def _comm_dir():
    """Per-user directory (mode 0700, owned by this uid) for the rendezvous files: not a predictable name in a
    world-writable place."""
    base = os.environ.get("BGP_COMM_DIR")
    if not base:
        root = os.environ.get("XDG_RUNTIME_DIR")
        if not (root and os.path.isdir(root) and os.access(root, os.W_OK)):
            root = tempfile.gettempdir()
        base = os.path.join(root, "bgp_comm_%d" % os.getuid())
    os.makedirs(base, mode=0o700, exist_ok=True)
    st = os.stat(base)
    if st.st_uid != os.getuid():
        raise RuntimeError(f"rendezvous directory {base} belongs to uid {st.st_uid}, not to this user")
    if st.st_mode & 0o077 and not os.environ.get("BGP_COMM_DIR"):
        os.chmod(base, 0o700)
    return base


def _job_prefix(ws):
    """Names of this rendezvous' files: job (launcher pid + start time, port, run id), the launcher's restart count (a
    worker restarted by torchrun --max-restarts keeps its parent) and this process's group-formation attempt (a second
    init_process_group in one process): no two attempts share a name."""
    tag = "%s_%s_%s_ws%d_r%s_a%d" % (os.environ.get("MASTER_PORT", "29500"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                    _job_id(), ws, os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), _state["attempt"])
    return os.path.join(_comm_dir(), "job_" + tag.replace("/", "_"))


def _write_private(path, data):
    tmp = "%s.%d.tmp" % (path, os.getpid())
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(data)
    os.replace(tmp, path)


def _read_owned(path):
    """Contents of a rendezvous file, or None while it does not exist; a file of another user is refused."""
    try:
        fd = os.open(path, os.O_RDONLY)
    except OSError:
        return None
    with os.fdopen(fd, "rb") as f:
        if os.fstat(f.fileno()).st_uid != os.getuid():
            raise RuntimeError(f"rendezvous file {path} is not owned by this user")
        return f.read()


def _cleanup_job_files(prefix):
    import glob

    for p in glob.glob(prefix + ".*"):
        try:
            os.remove(p)
        except OSError:
            pass


def _uid_digest(uid):
    import hashlib

    return hashlib.sha256(bytes(uid)).hexdigest()[:16].encode()


_T_START = time.time()


def _fresh_abort(prefix):
    """An abort marker written during THIS process's lifetime (a leftover of an earlier attempt under a reused name is
    older than this process and does not count)."""
    try:
        return os.stat(prefix + ".abort").st_mtime >= _T_START - 1.0
    except OSError:
        return False


def _exchange_unique_id_files(rank, ws, timeout, device_id=""):
    """Single node, files in the per-user directory, a handshake that no leftover can satisfy:
      1. every rank r > 0 writes <job>.hello.<r> = a nonce it has just drawn + "|" + the identity of its device (PCI bus id;
         and rewrites it should rank 0's initial clean-up remove it);
      2. rank 0 collects the hellos, refuses two ranks on ONE device (answer: nonce + "!" + message -- before librccl has been
         touched by anybody), else creates the id and answers every hello with <job>.ack.<r> = that nonce + the 128 id bytes; a
         rank only accepts an ack that carries ITS nonce -- written by a live rank 0 of this attempt, whatever files an earlier,
         killed attempt under the same name (a reused BGP_COMM_JOB) has left behind;
      3. every rank writes <job>.st.<r> = b"ok:<digest of the id it holds>:<identity of its device>" or b"fail: ..." and waits
         for all ws of them; a status only counts when it names the same id (the id is unique to the attempt).  Two ranks that
         name the SAME device (PCI bus id: a launcher that pins one GPU per rank through HIP_VISIBLE_DEVICES makes every rank see
         "device 0", and only the identities tell whether those are eight GPUs or one) are every rank's verdict "fail": RCCL
         refuses two ranks on one device, and it must not be found out inside ncclCommInitRank.
    Returns the id when ALL ranks hold it; raises RuntimeError (on every rank alike) otherwise -- before anyone has
    entered ncclCommInitRank, so one rank timing out cannot leave the others blocked in the collective."""
    from . import _lib

    prefix = _job_prefix(ws)
    _state["job_prefix"] = prefix
    deadline = time.monotonic() + timeout
    uid, err = None, None

    try:
        if rank == 0:
            _cleanup_job_files(prefix)  # (leftovers of a reused name go; a hello removed here is rewritten by its rank)
            import atexit

            atexit.register(_cleanup_job_files, prefix)
            # collect every rank's hello = nonce + "|" + the identity of its device ...
            hellos = {}
            while len(hellos) < ws - 1:
                for r in range(1, ws):
                    if r not in hellos:
                        buf = _read_owned("%s.hello.%d" % (prefix, r))
                        if buf is not None and len(buf) >= 33 and buf[32:33] == b"|":
                            hellos[r] = (buf[:32], buf[33:].decode(errors="replace"))
                if len(hellos) == ws - 1:
                    break
                if _fresh_abort(prefix):
                    raise RuntimeError("another rank gave the native group up")
                if time.monotonic() > deadline:
                    raise RuntimeError("no hello from rank(s) %s within %.0f s"
                                       % (sorted(set(range(1, ws)) - set(hellos)), timeout))
                time.sleep(0.01)
            # ... and look at the devices BEFORE librccl is touched by anybody: two ranks on one GPU are refused here, with a
            # message instead of an id (a process that has loaded the system's librccl and then falls back to torch's gloo group
            # carries two RCCL copies to its exit)
            named = sorted([(str(device_id), 0)] + [(dev, r) for r, (_n, dev) in hellos.items()])
            shared = [(a[1], b[1], a[0]) for a, b in zip(named, named[1:]) if a[0] and a[0] == b[0]]
            if shared:
                msg = "ranks %d and %d share device %s: RCCL needs a GPU per rank" % shared[0]
                for r, (nonce, _dev) in hellos.items():
                    _write_private("%s.ack.%d" % (prefix, r), nonce + b"!" + msg.encode())
                raise RuntimeError(msg)
            uid = _lib.comm_unique_id()
            for r, (nonce, _dev) in hellos.items():
                _write_private("%s.ack.%d" % (prefix, r), nonce + uid)
        else:
            nonce = os.urandom(16).hex().encode()
            hello = "%s.hello.%d" % (prefix, rank)
            mine_hello = nonce + b"|" + str(device_id).encode()
            _write_private(hello, mine_hello)
            while uid is None:
                buf = _read_owned("%s.ack.%d" % (prefix, rank))
                if buf is not None and buf[:32] == nonce and buf[32:33] == b"!":
                    raise RuntimeError(buf[33:].decode(errors="replace"))
                if buf is not None and buf[:32] == nonce and len(buf) == 32 + _lib.COMM_ID_BYTES:
                    uid = buf[32:]
                elif _fresh_abort(prefix):
                    raise RuntimeError("another rank gave the native group up")
                elif time.monotonic() > deadline:
                    raise RuntimeError(f"no ncclUniqueId from rank 0 ({prefix}.ack.{rank}) within {timeout:.0f} s")
                else:
                    if _read_owned(hello) != mine_hello:
                        _write_private(hello, mine_hello)
                    time.sleep(0.01)
    except Exception as exc:  # reported to the others below, then raised
        err = exc
    ok_head = b"ok:" + _uid_digest(uid) + b":" if err is None else None
    mine = ok_head + str(device_id).encode() if err is None else ("fail: %r" % (err,)).encode()
    _write_private("%s.st.%d" % (prefix, rank), mine)
    # the verdict of every rank
    bad = None if err is None else "rank %d: %r" % (rank, err)
    pending = set(range(ws))
    devices = {}
    while pending and bad is None:
        for r in sorted(pending):
            buf = _read_owned("%s.st.%d" % (prefix, r))
            if buf is None:
                continue
            if buf.startswith(b"ok:"):
                if buf.startswith(ok_head):
                    pending.discard(r)
                    devices[r] = buf[len(ok_head):].decode(errors="replace")
                # (another digest: a leftover of an earlier attempt -- wait for this attempt's status)
            elif buf.startswith(b"fail:") and os.stat("%s.st.%d" % (prefix, r)).st_mtime >= _T_START - 1.0:
                pending.discard(r)
                bad = "rank %d: %s" % (r, buf.decode(errors="replace"))
        if pending and bad is None:
            if _fresh_abort(prefix):
                bad = "another rank gave the native group up"
            elif time.monotonic() > deadline + 5.0:
                bad = "no status from rank(s) %s within %.0f s" % (sorted(pending), timeout)
            else:
                time.sleep(0.01)
    if bad is None:  # every rank holds the id: do they hold a device each?  (the same table on every rank: the same verdict)
        named = sorted((dev, r) for r, dev in devices.items() if dev)
        shared = [(a[1], b[1], a[0]) for a, b in zip(named, named[1:]) if a[0] == b[0]]
        if shared:
            bad = "ranks %d and %d share device %s: RCCL needs a GPU per rank" % shared[0]
    if bad is not None:
        try:  # a rank that arrives late must not walk into the collective alone
            _write_private(prefix + ".abort", bad.encode())
        except OSError:
            pass
        raise RuntimeError("native RCCL group not formed (%s)" % bad)
    return uid


def _exchange_unique_id_tcp(rank, ws, timeout):
    """Across nodes: rank 0 serves the id on MASTER_ADDR:(MASTER_PORT+1)."""
    from . import _lib

    host, port = _comm_endpoint()
    if rank == 0:
        uid = _lib.comm_unique_id()
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


def _exchange_unique_id(rank, ws, timeout=120.0, device_id=""):
    from . import _lib

    if ws == 1:
        return _lib.comm_unique_id()
    local = os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1") \
        and os.environ.get("BGP_COMM_TCP") != "1"
    if local:
        return _exchange_unique_id_files(rank, ws, timeout, device_id)
    return _exchange_unique_id_tcp(rank, ws, timeout)


_VISIBLE_DEVICES_VARS = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL")


def plan_group(ndev, local_rank, local_ws, comm_ok, name=None, device=None, pinned=None):
    """(backend to form, device ordinal of this rank) from what this rank sees, without any exchange -- every rank of a node
    sees the same facts, so every rank plans the same backend:

    * a backend asked for by name is taken (``rccl`` without a device raises: there is no CPU fallback);
    * at least as many visible devices as local ranks and a loadable librccl -> ``rccl``, device ``local_rank % ndev`` (the
      driver's ``torch.distributed.run --nproc-per-node 8`` on an 8-GPU node: rank r on GPU r);
    * FEWER visible devices than local ranks but a visible-devices variable in the environment (``HIP_VISIBLE_DEVICES`` and its
      relatives: a launcher that pins one GPU per rank makes every rank see ONE device, ordinal 0) -> ``rccl`` is tried with
      device ``local_rank % ndev``, and the rendezvous decides from the devices' PCI bus ids whether the ranks really hold a GPU
      each (``_exchange_unique_id_files``); a shared GPU is every rank's verdict and the group falls back to gloo;
    * otherwise (ranks sharing a device, or no device: the CPU tests) -> ``gloo``."""
    if name not in (None, "rccl", "gloo"):
        raise ValueError(f"unknown backend {name!r}: the GPU exchange is 'rccl' (libbgp's own communicator); 'gloo' is "
                         "the CPU group of the tests and of ranks that share a device")
    if pinned is None:
        pinned = any(os.environ.get(v) for v in _VISIBLE_DEVICES_VARS)
    if name is None:
        if ndev >= max(local_ws, 1) and comm_ok:
            name = "rccl"
        elif ndev >= 1 and comm_ok and pinned:
            name = "rccl"
        else:
            name = "gloo"
    if name == "gloo":
        return "gloo", None
    if ndev < 1:
        raise RuntimeError("BGP_DIST_BACKEND=rccl needs an MI355X (no CPU fallback); use gloo for CPU tests")
    return "rccl", ((local_rank % ndev) if device is None else int(device))


def init_process_group(backend=None, device=None):
    """Join the process group when launched with WORLD_SIZE > 1 (or BGP_DIST_FORCE=1).  Returns
    (rank, local_rank, world_size)."""
    rank, local_rank, ws = world()
    if _state["backend"] is not None or (ws <= 1 and os.environ.get("BGP_DIST_FORCE") != "1"):
        return rank, local_rank, ws
    from . import _lib

    _state["attempt"] += 1  # (every rank of a job forms its groups in the same order: the counters agree)

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    name = backend or os.environ.get("BGP_DIST_BACKEND") or None
    explicit = bool(name)
    ndev = _lib.device_count() if name != "gloo" else 0
    # decided from what every rank of a node sees alike, without any exchange (plan_group)
    name, dev = plan_group(ndev, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", ws)),
                           bool(ndev) and name != "gloo" and _lib.comm_available(), name, device)
    if name == "rccl":
        try:
            # (raises on EVERY rank when any rank failed, or when two ranks turn out to sit on one GPU: single node)
            try:
                ident = _lib.device_identity(dev)
            except Exception:
                ident = ""  # (no identity: the shared-device check is skipped for this rank)
            uid = _exchange_unique_id(rank, ws, device_id=ident)
            _state["comm"] = _lib.Comm(dev, rank, ws, uid)
        except Exception as exc:
            if explicit:
                raise
            print(f"[bayes_skopt_amd.distributed] rank {rank}: native RCCL group failed ({exc}); using gloo",
                  file=sys.stderr, flush=True)
            name, dev = "gloo", None
    if name == "gloo":
        import datetime

        dist = _torch_dist()
        if not dist.is_initialized():
            with _stdout_to_stderr():
                dist.init_process_group(backend="gloo", rank=rank, world_size=ws, timeout=datetime.timedelta(seconds=300))
                dist.barrier()  # (gloo connects its pairs lazily: make it talk now, while descriptor 1 is redirected)
    _state.update(backend=name, rank=rank, world=ws, device=dev)
    return rank, local_rank, ws


def destroy_process_group():
    if _state["backend"] == "rccl":
        _state["comm"].close()
    elif _state["backend"] is not None:
        dist = _torch_dist()
        if dist.is_initialized():
            dist.destroy_process_group()
    if _state["job_prefix"] and _state["rank"] == 0:
        _cleanup_job_files(_state["job_prefix"])  # (not left to atexit, which a SIGTERM skips)
    _state.update(backend=None, comm=None, rank=0, world=1, device=None, job_prefix=None)


def abort_process_group():
    """This rank is going down OUTSIDE a collective (KeyboardInterrupt, SystemExit): make the peers fail at once instead of
    leaving them in the all-gather until ``BGP_COMM_TIMEOUT_S`` -- ``ncclCommAbort`` on the native backend.  The gloo group of
    the CPU tests has no abort; its own time-out applies there."""
    if _state["backend"] == "rccl" and _state["comm"] is not None:
        try:
            _state["comm"].abort()
        except Exception:
            pass


def group_info(device=None):
    """What formed: backend, world size, the rank count RCCL itself reports (ncclCommCount; None over gloo) and the
    device index of every rank (all-gathered)."""
    info = {"backend": _state["backend"], "world": _state["world"], "rccl_nranks": None, "rank_devices": None}
    if _state["backend"] is None:
        return info
    if _state["backend"] == "rccl":
        info["rccl_nranks"] = _state["comm"].nranks()
    dev = _state["device"] if device is None else device
    info["rank_devices"] = [int(v) for v in _allgather(np.array([-1.0 if dev is None else float(dev)])).ravel()]
    return info


def _allgather(a):
    """(world,) + a.shape, rank-major, on every rank."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if _state["backend"] == "rccl":
        return _state["comm"].allgather(a)
    import torch

    dist = _torch_dist()
    t = torch.from_numpy(a.copy())
    out = [torch.empty_like(t) for _ in range(_state["world"])]
    dist.all_gather(out, t)
    return torch.stack(out, dim=0).numpy()


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
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_rows(B, rank, ws):
    """Row range [lo, hi) of a B-row proposal block owned by `rank` (contiguous, sizes differ by <= 1)."""
    return (B * rank) // ws, (B * (rank + 1)) // ws


def _assemble(parts, B, ws):
    full = np.empty(B)
    for r in range(ws):
        lo, hi = shard_rows(B, r, ws)
        full[lo:hi] = parts[r][: hi - lo]
    return full


def _raise_if_any_rank_failed(errs):
    bad = [(r, int(e)) for r, e in enumerate(errs) if int(e) != 0]
    if bad:
        raise ShardedEvaluationError("rank(s) %s failed in their share of a sharded log-probability block (codes %s); every "
                                     "rank stops" % ([r for r, _e in bad], [e for _r, e in bad]))


def allgather_lml(ctx, B, local=None, error=0):
    """The B log-likelihoods of a sharded proposal block on every rank.  Native group: ``ctx`` holds this rank's
    submitted rows (``Context.lml_submit``) and the values travel device to device (bgp_lml_batch_wait_allgather);
    gloo (CPU tests, ranks sharing a GPU): ``local`` = this rank's values, already collected on the host.
    ``error`` != 0: this rank's own share failed.  It STILL takes part -- a rank that stayed away would leave its peers in
    the collective for ever -- and sends the code in a status word next to its values; every rank then raises
    ``ShardedEvaluationError`` from the same gathered words."""
    ws = _state["world"]
    per = -(-B // ws)
    if _state["backend"] == "rccl" and local is None:
        vals, errs = ctx.lml_wait_allgather(_state["comm"], per, local_error=int(error))
    else:
        buf = np.full(per + 1, np.nan)
        if local is not None and len(local) and not error:
            buf[: len(local)] = local
        buf[per] = float(error)
        g = _allgather(buf)
        vals, errs = g[:, :per], g[:, per]
    _raise_if_any_rank_failed(errs)
    return _assemble(vals, B, ws)


def shard_log_prob(fn):
    """Wrap a vectorised log-probability ``fn(Theta (B,p), **kw) -> (B,)`` so that each rank evaluates only
    its own rows and the full vector is re-assembled with one all-gather of host values (exact single-ensemble sharding
    of ANY log-probability; BayesGPR's own device path uses the device-resident gather of ``allgather_lml``).
    Every rank must call it with the same Theta (same sampler RNG on every rank).  Outside a process
    group it is ``fn`` itself."""

    def wrapped(Theta, *args, **kwargs):
        Theta = np.atleast_2d(np.asarray(Theta, dtype=np.float64))
        if _state["backend"] is None:
            return fn(Theta, *args, **kwargs)
        ws, rank = _state["world"], _state["rank"]
        B = Theta.shape[0]
        lo, hi = shard_rows(B, rank, ws)
        local, failure = np.zeros(0), None
        try:
            if hi > lo:
                local = np.asarray(fn(Theta[lo:hi], *args, **kwargs), dtype=np.float64)
                if local.shape != (hi - lo,):
                    raise ValueError("log-probability returned shape %r for %d rows" % (local.shape, hi - lo))
        except Exception as exc:  # the peers are (about to be) in the collective: take part, say so, then raise
            failure = exc
        try:
            return allgather_lml(None, B, local=local, error=0 if failure is None else 1)
        except ShardedEvaluationError as err:
            if failure is not None:
                raise err from failure
            raise

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

    t = torch.from_numpy(arr.copy())
    _torch_dist().broadcast(t, src=src)
    return t.numpy()
