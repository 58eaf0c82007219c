"""ctypes binding of libbgp.so (the C-ABI declared in include/bgp.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C bayes-skopt_amd/csrc``.
There is NO CPU fallback: if the shared object is missing, or no gfx950 device is visible when
a context is created, the product path raises.
"""
import atexit
import weakref
import ctypes as C
import os

import numpy as np

# Kernel arguments in device memory instead of host memory (a HIP runtime switch, read when the runtime initialises: it only
# takes effect if nothing in the process has touched the GPU before this import; an explicit setting of the user's wins).  The
# launch schedule is a few dozen dependent launches per factorisation, each of which otherwise starts with a read of its
# arguments across PCIe: measured on MI355X (tools/persist_probe.py, bench.py) 0.666 -> 0.627 ms for 32 matrices of n = 1024,
# 2.22 -> 2.03 ms for one of n = 4096, 15.64 -> 15.47 ms per step at config C, results identical.
# It is a PROCESS-WIDE setting (every HIP user of the process and its children inherit it): BGP_NO_ENV_DEFAULTS=1 leaves the
# environment alone; `env_defaults()` says what this import did, bench.py records it in its line.
_ENV_DEFAULTS = {}
if os.environ.get("BGP_NO_ENV_DEFAULTS", "0") in ("", "0"):
    for _k, _v in (("HIP_FORCE_DEV_KERNARG", "1"),):
        if _k not in os.environ:
            os.environ[_k] = _v
            _ENV_DEFAULTS[_k] = _v


def env_defaults():
    """Environment variables this import set because the user had not (none with BGP_NO_ENV_DEFAULTS=1), and the value of the
    HIP runtime switch the process runs with."""
    return {"set_by_import": dict(_ENV_DEFAULTS), "HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbgp.so")

FORM = {"product": 0, "sum": 1}
STATIONARY = {"rbf": 0, "matern12": 1, "matern32": 2, "matern52": 3}


class BgpError(RuntimeError):
    pass


class NotPositiveDefinite(BgpError):
    """bgp_sample_y: predictive covariance (+ jitter) is not numerically positive definite."""


BGP_ERR_NOTPD = 5


class KernelSpecStruct(C.Structure):
    _fields_ = [("form", C.c_int), ("stationary", C.c_int), ("d", C.c_int)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes): every symbol include/bgp.h declares
SIGNATURES = {
    "bgp_device_count": (C.c_int, []),
    "bgp_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "bgp_last_error": (C.c_char_p, []),
    "bgp_version": (C.c_char_p, []),
    "bgp_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(KernelSpecStruct), C.c_int,
                                 C.POINTER(_vp)]),
    "bgp_ctx_update_data": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "bgp_ctx_destroy": (None, [_vp]),
    "bgp_lml_batch": (C.c_int, [_vp, C.c_int, _dp, _dp, _ip]),
    "bgp_lml_batch_submit": (C.c_int, [_vp, C.c_int, _dp]),
    "bgp_lml_batch_warped_submit": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "bgp_lml_batch_wait": (C.c_int, [_vp, _dp, _ip]),
    "bgp_lml_batch_warped": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _ip]),
    "bgp_ctx_set_warp": (C.c_int, [_vp, _dp]),
    "bgp_beta_cdf": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "bgp_lml_grad_batch": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _ip]),
    "bgp_kernel_matrix": (C.c_int, [_vp, _dp, _dp]),
    "bgp_posterior_batch": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip]),
    "bgp_predict_batch": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, _dp, _dp, _dp]),
    "bgp_acq_batch": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, C.c_double, C.c_double, C.c_int, _ip, _dp, C.c_int, _dp]),
    "bgp_acq_values": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_int, _ip, _dp, C.c_int, _dp]),
    "bgp_pvrs": (C.c_int, [_vp, _dp, C.c_int, _dp, C.c_int, _dp, _dp]),
    "bgp_pvrs_prepare": (C.c_int, [_vp, _dp, C.c_int, _ip]),
    "bgp_sample_y": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_double, _dp]),
    "bgp_sample_y_batch": (C.c_int, [_vp, C.c_int, _ip, _dp, C.c_int, _dp, _dp, C.c_double, _dp, _ip]),
    "bgp_lml_batch_gram": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, _ip]),
    "bgp_posterior_batch_gram": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, _dp, _dp, _dp, _ip]),
    "bgp_predict_batch_gram": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]),
    "bgp_comm_abort": (C.c_int, [_vp]),
    "bgp_mcmc_begin": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _ip, _dp, _ip, _dp, _dp, _dp]),
    "bgp_mcmc_begin_ex": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _dp, _ip, _dp, _dp, _dp]),
    "bgp_mcmc_progress": (C.c_int, [_vp, _ip]),
    "bgp_comm_init_loopback": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_longlong, C.POINTER(_vp)]),
    "bgp_comm_bench_lml_gather": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _dp]),
    "bgp_mcmc_steps": (C.c_int, [_vp, C.c_int, _ip, _ip, _dp, _dp, _dp]),
    "bgp_mcmc_end": (C.c_int, [_vp, _dp, _dp, _dp, _dp, C.POINTER(C.c_longlong), _ip]),
    "bgp_mcmc_run": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _ip, _dp, _ip, _dp, _dp, _dp, _ip, _ip, _dp, _dp, _dp, _dp, _dp, _dp,
                               _dp, C.POINTER(C.c_longlong), _ip]),
    "bgp_comm_available": (C.c_int, []),
    "bgp_comm_unique_id": (C.c_int, [_vp]),
    "bgp_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "bgp_comm_destroy": (None, [_vp]),
    "bgp_comm_allgather": (C.c_int, [_vp, _dp, C.c_size_t, _dp]),
    "bgp_comm_allreduce_max": (C.c_int, [_vp, _dp, C.c_size_t]),
    "bgp_comm_broadcast": (C.c_int, [_vp, _dp, C.c_size_t, C.c_int]),
    "bgp_comm_barrier": (C.c_int, [_vp]),
    "bgp_comm_nranks": (C.c_int, [_vp, _ip]),
    "bgp_lml_batch_wait_allgather": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _dp, _ip]),
    "bgp_device_synchronize": (C.c_int, [C.c_int]),
    "bgp_set_streams": (C.c_int, [_vp, C.c_int]),
    "bgp_last_timing": (C.c_int, [_vp, _dp, _ip]),
    "bgp_set_timing": (C.c_int, [_vp, C.c_int]),
    "bgp_set_persist": (C.c_int, [_vp, C.c_int]),
    "bgp_last_timing_columns": (C.c_int, [_vp, _dp, _ip]),
    "bgp_debug_workspace": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "bgp_debug_cov_factor": (C.c_int, [_vp, _ip, _dp]),
    "bgp_persist_stats": (C.c_int, [_vp, C.POINTER(C.c_longlong)]),
    "bgp_lml_gen_stats": (C.c_int, [_vp, C.POINTER(C.c_longlong)]),
    "bgp_debug_ps_trace": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.c_size_t]),
    "bgp_debug_pivot_root": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp]),
    "bgp_bench_mfma_f64": (C.c_int, [C.c_int, C.c_int, _dp]),
    "bgp_bench_hbm_copy": (C.c_int, [C.c_int, C.c_longlong, C.c_int, _dp]),
    "bgp_mfma_f64_layout": (C.c_int, [C.c_int, _ip, _ip]),
}

_lib = None


def load():
    """Load libbgp.so (once) and declare every prototype.  Raises BgpError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BgpError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C bayes-skopt_amd/csrc`). "
            "There is no CPU fallback."
        )
    # multi-process GPU work on this platform needs dmabuf IPC (RCCL's ncclCommInitRank fails with
    # "hipIpcGetMemHandle: invalid argument" otherwise); the runtime reads the switch when it initialises, i.e. at the
    # first HIP call of the process, so it is set before the library is even loaded
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise BgpError(f"{what} failed (code {rc}): {load().bgp_last_error().decode()}")


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(_dp if a.dtype == np.float64 else _ip)


def device_count():
    return int(load().bgp_device_count())


def device_identity(device=0):
    """PCI bus id of a visible device (``bgp_device_pci_bus_id``): what tells ranks that all see "device 0" apart."""
    buf = C.create_string_buffer(64)
    _check(load().bgp_device_pci_bus_id(int(device), buf, 64), "bgp_device_pci_bus_id")
    return buf.value.decode()


ACQ_EI, ACQ_MEAN, ACQ_LCB, ACQ_STD = 0, 1, 2, 3  # include/bgp.h BGP_ACQ_*
ACQ_MAX = 8


_live_contexts = weakref.WeakSet()


@atexit.register
def _close_live_contexts():
    # contexts still alive when the interpreter goes down are closed while the HIP runtime is still there (the last one
    # takes the library's process-wide streams with it: left alive they crashed a profiler's teardown)
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass


class Context:
    """Owns one device context: training set resident in HBM + the batched-Cholesky workspace."""

    def __init__(self, X, y, alpha_diag, form="product", stationary="matern52", max_batch=64, device=0):
        lib = load()
        X = _c(np.atleast_2d(X))
        n, d = X.shape
        y = _c(y).reshape(n)
        alpha_diag = _c(np.broadcast_to(np.asarray(alpha_diag, dtype=np.float64), (n,)))
        self.n, self.d, self.p = n, d, d + 2
        self.form, self.stationary = form, stationary
        self.max_batch = int(max_batch)
        self.device = int(device)
        spec = KernelSpecStruct(FORM[form], STATIONARY[stationary], d)
        h = _vp()
        if device_count() <= 0:
            raise BgpError("no HIP device visible: the BayesGPR hot path needs an MI355X (no CPU fallback)")
        _check(lib.bgp_ctx_create(self.device, n, d, _p(X), _p(y), _p(alpha_diag), C.byref(spec), self.max_batch,
                                  C.byref(h)), "bgp_ctx_create")
        self._h = h
        self._lib = lib
        _live_contexts.add(self)
        self._pending, self._pending_H = 0, None  # batch handed to lml_submit and not yet collected
        self._timing = False
        # canonical vectors of the posteriors whose K^-1 / alpha are resident on the device
        # (None after a call that overwrites them: gradient / pvrs_prepare / update_data)
        self.resident_H = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bgp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update_data(self, X, y, alpha_diag):
        X = _c(np.atleast_2d(X))
        n, d = X.shape
        if d != self.d:
            raise ValueError("input dimension changed")
        y = _c(y).reshape(n)
        alpha_diag = _c(np.broadcast_to(np.asarray(alpha_diag, dtype=np.float64), (n,)))
        _check(self._lib.bgp_ctx_update_data(self._h, n, _p(X), _p(y), _p(alpha_diag)), "bgp_ctx_update_data")
        self.n = n
        self.resident_H = None

    def _H(self, H):
        H = _c(np.atleast_2d(H))
        if H.shape[1] != self.p:
            raise ValueError(f"canonical hyper-parameter vectors must have length d+2={self.p}, got {H.shape[1]}")
        return H

    def lml(self, H, return_status=False):
        H = self._H(H)
        B = H.shape[0]
        out = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        _check(self._lib.bgp_lml_batch(self._h, B, _p(H), _p(out), _p(st)), "bgp_lml_batch")
        return (out, st) if return_status else out

    def lml_submit(self, H):
        """Enqueue ``lml(H)`` on the device and return at once (False when the batch cannot go asynchronously: larger
        than max_batch, or per-launch timing on); ``lml_wait()`` collects the result."""
        H = self._H(H)
        if H.shape[0] > self.max_batch or H.shape[0] == 0 or self._timing:
            return False  # (per-launch timing synchronises inside the call: the caller uses the synchronous form)
        _check(self._lib.bgp_lml_batch_submit(self._h, H.shape[0], _p(H)), "bgp_lml_batch_submit")
        self._pending, self._pending_H = H.shape[0], H  # (H stays alive until the upload has certainly happened)
        return True

    def mcmc_begin(self, coords, logp, nsteps, h_src, h_fixed, prior_kind, prior_par, comm=None, nwarp=0):
        """``bgp_mcmc_begin_ex``: open a device-resident run of the ensemble sampler (``nsteps`` steps of the (W, p) ensemble
        ``coords`` with log-probabilities ``logp``); the plan follows in segments through ``mcmc_steps``, ``mcmc_end`` collects.
        ``comm``: a ``Comm`` whose ranks share every half-step's proposal block (the sharded ensemble; every rank calls with the
        same arguments); ``nwarp`` = 2 d: the walkers' last 2 d entries are their own input-warp parameters."""
        coords = _c(np.asarray(coords, dtype=np.float64))
        W, p = coords.shape
        logp = _c(np.asarray(logp, dtype=np.float64))
        h_src = np.ascontiguousarray(h_src, dtype=np.int32)
        h_fixed = _c(np.asarray(h_fixed, dtype=np.float64))
        prior_kind = np.ascontiguousarray(prior_kind, dtype=np.int32)
        prior_par = _c(np.asarray(prior_par, dtype=np.float64))
        if h_src.shape != (self.d + 2,) or h_fixed.shape != (self.d + 2,) or prior_kind.shape != (p,) or prior_par.shape != (p, 5):
            raise ValueError("canonical map / prior tables have the wrong shape")
        _check(self._lib.bgp_mcmc_begin_ex(self._h, comm._h if comm is not None else None, int(nwarp), W, p, int(nsteps), _p(h_src),
                                           _p(h_fixed), _p(prior_kind), _p(prior_par), _p(coords), _p(logp)), "bgp_mcmc_begin")
        self._mcmc = (W, p, int(nsteps))

    def mcmc_progress(self):
        """Steps of the open run the device has worked through (``bgp_mcmc_progress``; never blocks)."""
        v = C.c_int(0)
        _check(self._lib.bgp_mcmc_progress(self._h, C.byref(v)), "bgp_mcmc_progress")
        return int(v.value)

    def mcmc_steps(self, plan):
        """``bgp_mcmc_steps``: hand over the next segment of the plan -- (movers, partners, zz, factors, logu), each
        (2 * nseg, W / 2) -- and return at once; the device works through it while the caller draws the next segment."""
        W, p, _ = self._mcmc
        movers, partners, zz, factors, logu = plan
        movers = np.ascontiguousarray(movers, dtype=np.int32)
        partners = np.ascontiguousarray(partners, dtype=np.int32)
        zz, factors, logu = (_c(np.asarray(a, dtype=np.float64)) for a in (zz, factors, logu))
        nhalf = movers.shape[0]
        if nhalf % 2 or any(a.shape != (nhalf, W // 2) for a in (movers, partners, zz, factors, logu)):
            raise ValueError("a plan segment must hold (2 * nseg, W / 2) rows of every array")
        try:
            _check(self._lib.bgp_mcmc_steps(self._h, nhalf // 2, _p(movers), _p(partners), _p(zz), _p(factors), _p(logu)), "bgp_mcmc_steps")
        except BaseException:
            self.mcmc_abandon()
            raise

    def mcmc_abandon(self):
        """Drop an open run (an exception between ``mcmc_begin`` and ``mcmc_end``): ``bgp_mcmc_end`` with nothing to
        collect closes it on the C side."""
        if getattr(self, "_mcmc", None) is not None:
            self._mcmc = None
            self._lib.bgp_mcmc_end(self._h, None, None, None, None, None, None)

    def mcmc_end(self):
        """``bgp_mcmc_end``: wait for the run and collect chain (nsteps, W, p), logp (nsteps, W), the final ensemble and its
        log-probabilities, accept counts (W,) and the four info words (non-finite proposal seen; run redone on the launch
        schedule; half-step of the first non-finite proposal; whether it was a NaN)."""
        W, p, nsteps = self._mcmc
        self._mcmc = None
        chain = np.empty((nsteps, W, p))
        lps = np.empty((nsteps, W))
        cout, lout = np.empty((W, p)), np.empty(W)
        nacc = np.zeros(W, dtype=np.int64)
        info = np.zeros(4, dtype=np.int32)
        _check(self._lib.bgp_mcmc_end(self._h, _p(chain), _p(lps), _p(cout), _p(lout), nacc.ctypes.data_as(C.POINTER(C.c_longlong)),
                                      _p(info)), "bgp_mcmc_end")
        return chain, lps, cout, lout, nacc, info

    def mcmc_run(self, coords, logp, plan, h_src, h_fixed, prior_kind, prior_par, comm=None, nwarp=0):
        """``bgp_mcmc_run``: the same run with the whole plan handed over at once."""
        self.mcmc_begin(coords, logp, np.asarray(plan[0]).shape[0] // 2, h_src, h_fixed, prior_kind, prior_par, comm=comm, nwarp=nwarp)
        self.mcmc_steps(plan)
        return self.mcmc_end()

    def lml_warped_submit(self, H, W):
        """Asynchronous ``lml_warped``: False when the batch cannot go asynchronously (see ``lml_submit``)."""
        H = self._H(H)
        W = _c(np.atleast_2d(W))
        B = H.shape[0]
        if W.shape != (B, 2 * self.d):
            raise ValueError(f"warp parameters must be (B, 2d) = ({B}, {2 * self.d}), got {W.shape}")
        if B > self.max_batch or B == 0 or self._timing:
            return False
        _check(self._lib.bgp_lml_batch_warped_submit(self._h, B, _p(H), _p(W)), "bgp_lml_batch_warped_submit")
        self._pending, self._pending_H = B, (H, W)
        return True

    def lml_wait(self, return_status=False):
        B = self._pending  # (0: the C layer answers BGP_ERR_STATE, "nothing submitted")
        self._pending, self._pending_H = 0, None
        out = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        _check(self._lib.bgp_lml_batch_wait(self._h, _p(out), _p(st)), "bgp_lml_batch_wait")
        return (out, st) if return_status else out

    def lml_warped(self, H, W, return_status=False):
        """Per-walker warp: W is (B, 2d) log-space Beta parameters [wa_1..wa_d, wb_1..wb_d]."""
        H = self._H(H)
        W = _c(np.atleast_2d(W))
        B = H.shape[0]
        if W.shape != (B, 2 * self.d):
            raise ValueError(f"warp parameters must be (B, 2d) = ({B}, {2 * self.d}), got {W.shape}")
        out = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        _check(self._lib.bgp_lml_batch_warped(self._h, B, _p(H), _p(W), _p(out), _p(st)), "bgp_lml_batch_warped")
        return (out, st) if return_status else out

    def set_warp(self, w):
        """Context-level warp (2d log-space parameters) or None to clear."""
        self.resident_H = None
        if w is None:
            _check(self._lib.bgp_ctx_set_warp(self._h, C.cast(None, _dp)), "bgp_ctx_set_warp")
            return
        w = _c(w).reshape(2 * self.d)
        _check(self._lib.bgp_ctx_set_warp(self._h, _p(w)), "bgp_ctx_set_warp")

    def beta_cdf(self, X, w):
        X = _c(np.atleast_2d(X))
        w = _c(w).reshape(2 * self.d)
        out = np.empty_like(X)
        _check(self._lib.bgp_beta_cdf(self._h, X.shape[0], _p(X), _p(w), _p(out)), "bgp_beta_cdf")
        return out

    def lml_grad(self, H):
        H = self._H(H)
        B = H.shape[0]
        out = np.empty(B)
        grad = np.empty((B, self.p))
        st = np.zeros(B, dtype=np.int32)
        self.resident_H = None
        _check(self._lib.bgp_lml_grad_batch(self._h, B, _p(H), _p(out), _p(grad), _p(st)), "bgp_lml_grad_batch")
        return out, grad, st

    def kernel_matrix(self, h):
        H = self._H(h)
        K = np.empty((self.n, self.n))
        _check(self._lib.bgp_kernel_matrix(self._h, _p(H), _p(K)), "bgp_kernel_matrix")
        return K

    def posterior(self, H, want_L=False, want_alpha=True, want_K_inv=False):
        H = self._H(H)
        B, n = H.shape[0], self.n
        L = np.empty((B, n, n)) if want_L else None
        a = np.empty((B, n)) if want_alpha else None
        Ki = np.empty((B, n, n)) if want_K_inv else None
        lml = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        nul = C.cast(None, _dp)
        self.resident_H = None
        _check(self._lib.bgp_posterior_batch(self._h, B, _p(H), _p(L) if want_L else nul,
                                             _p(a) if want_alpha else nul, _p(Ki) if want_K_inv else nul, _p(lml),
                                             _p(st)), "bgp_posterior_batch")
        if np.all(st == 0):
            self.resident_H = H.copy()
        return {"L": L, "alpha": a, "K_inv": Ki, "lml": lml, "status": st}

    # ---- generic kernel expression trees: host-evaluated kernel matrices in, device arithmetic behind them
    def _Kstack(self, K):
        K = _c(K)
        if K.ndim == 2:
            K = K[None, :, :]
        if K.ndim != 3 or K.shape[1:] != (self.n, self.n):
            raise ValueError(f"kernel matrices must be (B, {self.n}, {self.n}), got {K.shape}")
        return K

    def lml_gram(self, K, use_alpha=True, return_status=False):
        """Log-marginal likelihood of B host-evaluated kernel matrices ``kernel_(X_train)`` (alpha added on the device)."""
        K = self._Kstack(K)
        B = K.shape[0]
        lml = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        _check(self._lib.bgp_lml_batch_gram(self._h, B, _p(K), int(bool(use_alpha)), _p(lml), _p(st)), "bgp_lml_batch_gram")
        return (lml, st) if return_status else lml

    def posterior_gram(self, K, use_alpha=True, want_L=False, want_alpha=True, want_K_inv=False):
        K = self._Kstack(K)
        B, n = K.shape[0], self.n
        L = np.empty((B, n, n)) if want_L else None
        a = np.empty((B, n)) if want_alpha else None
        Ki = np.empty((B, n, n)) if want_K_inv else None
        lml = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        nul = C.cast(None, _dp)
        self.resident_H = None  # (the resident posteriors belong to no canonical vector)
        _check(self._lib.bgp_posterior_batch_gram(self._h, B, _p(K), int(bool(use_alpha)), _p(L) if want_L else nul,
                                                  _p(a) if want_alpha else nul, _p(Ki) if want_K_inv else nul, _p(lml),
                                                  _p(st)), "bgp_posterior_batch_gram")
        return {"L": L, "alpha": a, "K_inv": Ki, "lml": lml, "status": st}

    def predict_gram(self, Ks, kss, Kss=None):
        """Predict for the B resident posteriors from host-evaluated ``Ks`` (B, m, n) = kernel_(Xq, X_train),
        ``kss`` (B, m) = kernel_.diag(Xq) and, for the full covariance, ``Kss`` (B, m, m) = kernel_(Xq)."""
        Ks = _c(Ks)
        if Ks.ndim == 2:
            Ks = Ks[None]
        B, m, n = Ks.shape
        if n != self.n:
            raise ValueError(f"Ks must have {self.n} columns, got {n}")
        kss = _c(np.broadcast_to(np.asarray(kss, dtype=np.float64), (B, m)))
        mean = np.empty((B, m))
        var = np.empty((B, m))
        nul = C.cast(None, _dp)
        cov = None
        if Kss is not None:
            Kss = _c(np.broadcast_to(np.asarray(Kss, dtype=np.float64), (B, m, m)))
            cov = np.empty((B, m, m))
        _check(self._lib.bgp_predict_batch_gram(self._h, B, m, _p(Ks), _p(kss), _p(Kss) if Kss is not None else nul,
                                                _p(mean), _p(var), _p(cov) if cov is not None else nul),
               "bgp_predict_batch_gram")
        return (mean, var, cov) if cov is not None else (mean, var)

    def has_pending(self):
        """A batch handed to ``lml_submit`` has not been collected yet."""
        return bool(self._pending)

    def predict(self, H_kernel, Xq, return_cov=False):
        H = self._H(H_kernel)
        B = H.shape[0]
        Xq = _c(np.atleast_2d(Xq))
        m = Xq.shape[0]
        mean = np.empty((B, m))
        var = np.empty((B, m))
        cov = np.empty((B, m, m)) if return_cov else None
        nul = C.cast(None, _dp)
        _check(self._lib.bgp_predict_batch(self._h, B, _p(H), m, _p(Xq), _p(mean), _p(var),
                                           _p(cov) if return_cov else nul), "bgp_predict_batch")
        return (mean, var, cov) if return_cov else (mean, var)

    def acq(self, H_kernel, Xq, y_mean, y_std, kinds, params, n_samples):
        """Closed-form acquisitions (ACQ_EI / ACQ_MEAN / ACQ_LCB / ACQ_STD) of the resident posteriors at Xq, averaged
        over the draws on the device (bgp_acq_batch): (len(kinds), m)."""
        H = self._H(H_kernel)
        Xq = _c(np.atleast_2d(Xq))
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        params = _c(np.asarray(params, dtype=np.float64))
        out = np.empty((len(kinds), Xq.shape[0]))
        _check(self._lib.bgp_acq_batch(self._h, H.shape[0], _p(H), Xq.shape[0], _p(Xq), float(y_mean), float(y_std),
                                       len(kinds), _p(kinds), _p(params), int(n_samples), _p(out)), "bgp_acq_batch")
        return out

    def acq_values(self, mu, std, kinds, params, n_samples):
        """The same closed forms on given (B, m) mean / standard deviation rows (bgp_acq_values)."""
        mu = _c(np.atleast_2d(mu))
        std = _c(np.atleast_2d(std))
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        params = _c(np.asarray(params, dtype=np.float64))
        out = np.empty((len(kinds), mu.shape[1]))
        _check(self._lib.bgp_acq_values(self._h, mu.shape[0], mu.shape[1], _p(mu), _p(std), len(kinds), _p(kinds),
                                        _p(params), int(n_samples), _p(out)), "bgp_acq_values")
        return out

    def pvrs_prepare(self, h_kernel, has_alpha_vec):
        H = self._H(h_kernel)
        st = np.zeros(1, dtype=np.int32)
        self.resident_H = None
        _check(self._lib.bgp_pvrs_prepare(self._h, _p(H), int(bool(has_alpha_vec)), _p(st)), "bgp_pvrs_prepare")
        return int(st[0])

    def pvrs(self, h_kernel, Xcand, Xthompson):
        H = self._H(h_kernel)
        Xc = _c(np.atleast_2d(Xcand))
        Xt = _c(np.atleast_2d(Xthompson))
        covs = np.empty(Xc.shape[0])
        _check(self._lib.bgp_pvrs(self._h, _p(H), Xc.shape[0], _p(Xc), Xt.shape[0], _p(Xt), _p(covs)), "bgp_pvrs")
        return covs

    def sample_y(self, b, h_kernel, Xq, z, jitter=0.0):
        H = self._H(h_kernel)
        Xq = _c(np.atleast_2d(Xq))
        z = _c(np.atleast_2d(z))
        m = Xq.shape[0]
        if z.shape[1] != m:
            raise ValueError("z must be (n_draws, m)")
        out = np.empty_like(z)
        rc = self._lib.bgp_sample_y(self._h, int(b), _p(H), m, _p(Xq), z.shape[0], _p(z), float(jitter), _p(out))
        if rc == BGP_ERR_NOTPD:
            raise NotPositiveDefinite(self._lib.bgp_last_error().decode())
        _check(rc, "bgp_sample_y")
        return out

    def sample_y_batch(self, pidx, H_kernel, Xq, z, jitter=0.0):
        """One draw per resident posterior pidx[i]: returns (out (B, m), status (B,))."""
        H = self._H(H_kernel)
        Xq = _c(np.atleast_2d(Xq))
        z = _c(np.atleast_2d(z))
        pidx = np.ascontiguousarray(pidx, dtype=np.int32)
        B, m = H.shape[0], Xq.shape[0]
        if z.shape != (B, m) or pidx.shape != (B,):
            raise ValueError("z must be (B, m) and pidx (B,)")
        out = np.empty_like(z)
        st = np.zeros(B, dtype=np.int32)
        _check(self._lib.bgp_sample_y_batch(self._h, B, _p(pidx), _p(H), m, _p(Xq), _p(z), float(jitter), _p(out),
                                            _p(st)), "bgp_sample_y_batch")
        return out, st

    def debug_workspace(self, b):
        """(L, z) of batch slot b as the last ``lml`` call left them: the (npad, npad) working matrix (factor in its lower
        triangle) and the working right-hand side."""
        npad = -(-self.n // 128) * 128
        L = np.empty((npad, npad))
        z = np.empty(npad)
        _check(self._lib.bgp_debug_workspace(self._h, int(b), _p(L), _p(z)), "bgp_debug_workspace")
        return L, z

    def debug_cov_factor(self):
        """The (mpad, mpad) factor of the predictive covariance the last ``sample_y`` left (lower triangle); None without one."""
        mp = C.c_int(0)
        _check(self._lib.bgp_debug_cov_factor(self._h, C.byref(mp), C.cast(None, _dp)), "bgp_debug_cov_factor")
        if mp.value == 0:
            return None
        L = np.empty((mp.value, mp.value))
        _check(self._lib.bgp_debug_cov_factor(self._h, C.byref(mp), _p(L)), "bgp_debug_cov_factor")
        return L

    def persist_stats(self):
        """Launch-free path bookkeeping (bgp_persist_stats): calls enqueued, time-outs, whether a time-out keeps the path off
        and for how many eligible calls."""
        v = (C.c_longlong * 4)()
        _check(self._lib.bgp_persist_stats(self._h, v), "bgp_persist_stats")
        return {"calls": int(v[0]), "timeouts": int(v[1]), "disabled": bool(v[2]), "cooldown_left": int(v[3])}

    def gen_stats(self):
        """LML batches whose kernel-matrix blocks (all but block column 0) were generated inside the trailing update of the first
        panel group, and the generating launches among their updates (bgp_lml_gen_stats)."""
        v = (C.c_longlong * 2)()
        _check(self._lib.bgp_lml_gen_stats(self._h, v), "bgp_lml_gen_stats")
        return {"batches": int(v[0]), "launches": int(v[1])}

    def ps_trace(self):
        """In-kernel timeline of the last launch-free call (needs BGP_PS_TRACE=1 at context creation): (chain, tile) --
        wall-clock stamps (100 MHz), 8 per (matrix, block column) of the chain role and 8 per tile task; None without a
        trace (bgp_debug_ps_trace)."""
        dims = (C.c_int * 3)()
        _check(self._lib.bgp_debug_ps_trace(self._h, dims, None, 0), "bgp_debug_ps_trace")
        B, nblk, total = dims[0], dims[1], dims[2]
        if B * nblk == 0:
            return None
        buf = np.zeros(B * nblk * 8 + total * 8, dtype=np.uint64)
        _check(self._lib.bgp_debug_ps_trace(self._h, dims, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size),
               "bgp_debug_ps_trace")
        return buf[: B * nblk * 8].reshape(B, nblk, 8), buf[B * nblk * 8:].reshape(total, 8)

    def set_streams(self, nstreams):
        _check(self._lib.bgp_set_streams(self._h, int(nstreams)), "bgp_set_streams")

    def set_persist(self, mode):
        """Launch-free factorisation of small batches: 1 on, 0 off, -1 as BGP_PERSIST says (bgp_set_persist)."""
        _check(self._lib.bgp_set_persist(self._h, int(mode)), "bgp_set_persist")

    def set_timing(self, enable):
        _check(self._lib.bgp_set_timing(self._h, int(bool(enable))), "bgp_set_timing")
        self._timing = bool(enable)

    def lml_wait_allgather(self, comm, per_rank, local_error=0):
        """Collective form of ``lml_wait`` for the exact single-ensemble sharding: every rank has submitted its own rows
        (possibly none) of the half-step's block; returns ((world, per_rank) log-likelihoods of all ranks, (world,) status
        words), gathered device to device over RCCL out of the contexts' resident result vectors
        (bgp_lml_batch_wait_allgather).  ``local_error`` > 0: this rank's own work failed -- it still takes part (its peers
        would block in the collective otherwise) and every rank finds the code in the status words."""
        out = np.empty((comm.world, int(per_rank)))
        errs = np.zeros(comm.world, dtype=np.int32)
        try:
            rc = self._lib.bgp_lml_batch_wait_allgather(self._h, comm._h, int(per_rank), int(local_error), _p(out), _p(errs))
        finally:
            self._pending, self._pending_H = 0, None  # (the C call consumes the pending batch whatever its outcome)
        _check(rc, "bgp_lml_batch_wait_allgather")
        return out, errs

    def last_timing(self):
        ms = np.zeros(5)
        cnt = np.zeros(4, dtype=np.int32)
        _check(self._lib.bgp_last_timing(self._h, _p(ms), _p(cnt)), "bgp_last_timing")
        names = ("kbuild", "potrf", "trsm", "syrk")
        out = {k: {"ms": float(ms[i]), "launches": int(cnt[i])} for i, k in enumerate(names)}
        out["device_total_ms"] = float(ms[4])
        cms, cn = C.c_double(0.0), C.c_int(0)
        _check(self._lib.bgp_last_timing_columns(self._h, C.byref(cms), C.byref(cn)), "bgp_last_timing_columns")
        # the look-ahead column launches inside "syrk" (K = 128 .. 128 (P-1) on one block column)
        out["syrk_columns"] = {"ms": float(cms.value), "launches": int(cn.value)}
        return out


COMM_ID_BYTES = 128


def comm_available(load_it=False):
    """True when librccl.so is there for the native multi-GPU backend.  By default the library is only LOOKED for (the paths
    bgp_comm.hip tries): loading the system's librccl into a process that may still fall back to torch's gloo group -- torch
    brings an RCCL of its own -- ends in a double free at exit.  ``load_it=True`` really dlopens it (bgp_comm_available)."""
    if load_it:
        return bool(load().bgp_comm_available())
    import ctypes.util

    return any(os.path.exists(p) for p in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so")) \
        or ctypes.util.find_library("rccl") is not None


def comm_unique_id():
    """ncclGetUniqueId through libbgp (rank 0): 128 opaque bytes every rank needs for ``Comm``."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().bgp_comm_unique_id(C.cast(buf, _vp)), "bgp_comm_unique_id")
    return buf.raw


class Comm:
    """RCCL communicator of this process (one process per GPU) behind the C-ABI: host arrays in, host arrays out."""

    def __init__(self, device, rank, world, unique_id):
        self._lib = load()
        self.rank, self.world = int(rank), int(world)
        h = _vp()
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        _check(self._lib.bgp_comm_init(int(device), self.rank, self.world, C.cast(buf, _vp), C.byref(h)), "bgp_comm_init")
        self._h = h

    @classmethod
    def loopback(cls, device, rank, world, key):
        """Loop-back communicator (``bgp_comm_init_loopback``): rank ``rank`` of ``world`` communicators of THIS process on one
        device, one per host thread; serves the in-stream exchange of a sharded resident sampler run only (tests)."""
        self = cls.__new__(cls)
        self._lib = load()
        self.rank, self.world = int(rank), int(world)
        h = _vp()
        _check(self._lib.bgp_comm_init_loopback(int(device), self.rank, self.world, int(key), C.byref(h)), "bgp_comm_init_loopback")
        self._h = h
        return self

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bgp_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allgather(self, a):
        a = _c(a)
        out = np.empty((self.world,) + a.shape)
        _check(self._lib.bgp_comm_allgather(self._h, _p(a), a.size, _p(out)), "bgp_comm_allgather")
        return out

    def allreduce_max(self, a):
        a = _c(np.array(a, dtype=np.float64, copy=True))
        _check(self._lib.bgp_comm_allreduce_max(self._h, _p(a), a.size), "bgp_comm_allreduce_max")
        return a

    def broadcast(self, a, root=0):
        a = _c(np.array(a, dtype=np.float64, copy=True))
        _check(self._lib.bgp_comm_broadcast(self._h, _p(a), a.size, int(root)), "bgp_comm_broadcast")
        return a

    def barrier(self):
        _check(self._lib.bgp_comm_barrier(self._h), "bgp_comm_barrier")

    def abort(self):
        """ncclCommAbort: the peers' collectives fail at once instead of waiting for this rank."""
        if getattr(self, "_h", None):
            self._lib.bgp_comm_abort(self._h)

    def bench_lml_gather(self, ctx, per, reps=200):
        """ms per round of the sharded resident sampler's in-stream exchange (``bgp_comm_bench_lml_gather``; every rank calls)."""
        v = C.c_double(0.0)
        _check(self._lib.bgp_comm_bench_lml_gather(ctx._h, self._h, int(per), int(reps), C.byref(v)), "bgp_comm_bench_lml_gather")
        return float(v.value)

    def nranks(self):
        """Ranks RCCL itself counts in the communicator (ncclCommCount)."""
        v = C.c_int(0)
        _check(self._lib.bgp_comm_nranks(self._h, C.byref(v)), "bgp_comm_nranks")
        return int(v.value)


def device_synchronize(device=0):
    _check(load().bgp_device_synchronize(int(device)), "bgp_device_synchronize")


def bench_mfma_f64(device=0, iters=20000):
    v = C.c_double(0.0)
    _check(load().bgp_bench_mfma_f64(device, iters, C.byref(v)), "bgp_bench_mfma_f64")
    return v.value


def bench_hbm_copy(device=0, nbytes=1 << 30, iters=10):
    v = C.c_double(0.0)
    _check(load().bgp_bench_hbm_copy(device, nbytes, iters, C.byref(v)), "bgp_bench_hbm_copy")
    return v.value


def pivot_root(x, device=0):
    """(sqrt(x), 1 / sqrt(x)) as the diagonal-block factorisation forms its pivots (bgp_debug_pivot_root)."""
    x = _c(x).ravel()
    s, r = np.empty_like(x), np.empty_like(x)
    _check(load().bgp_debug_pivot_root(int(device), x.size, _p(x), _p(s), _p(r)), "bgp_debug_pivot_root")
    return s, r


def mfma_f64_layout(device=0):
    rows = np.zeros(256, dtype=np.int32)
    cols = np.zeros(256, dtype=np.int32)
    _check(load().bgp_mfma_f64_layout(device, _p(rows), _p(cols)), "bgp_mfma_f64_layout")
    return rows.reshape(64, 4), cols.reshape(64, 4)
