"""Affine-invariant ensemble sampler (Goodman & Weare stretch move, red-blue split) -- the host
driver of the MCMC loop, written from scratch.

The reference constructs ``emcee.EnsembleSampler`` (3.1.6, absent from this image) at
``bask/bayesgpr.py:510-517``, seeds it at ``:518-521``, runs it at ``:522-524`` and reads the chain
at ``:528-530``.  This class keeps that interface (``run_mcmc``, ``get_chain``, ``random_state``
setter, ``acceptance_fraction``) and consumes the numpy ``RandomState`` in emcee's published
order: per step one move-selection draw and a shuffle of the red/blue
labels; per half-step ``rand(Ns)`` (stretch factors), ``randint(Nc, size=Ns)`` (partners) and Ns
uniform draws (accept tests).

What is MI355X-specific: ``log_prob_fn`` is called ONCE per half-step with the whole (Ns, ndim)
block of proposals -- the data-parallel axis of the hot path -- so that all Ns
kernel-matrix builds + Cholesky factorisations of a half-step run as one batch on the GPU.  The
accept/reject bookkeeping (O(W) scalar work per half-step) stays on the host.
"""
import sys
import time

import numpy as np

__all__ = ["EnsembleSampler", "walkers_independent"]

_told = set()


def _tell_once(reason):
    """One line on stderr, once per process and reason: a run that could have stayed on the device is driven from the host."""
    if reason and reason not in _told:
        _told.add(reason)
        print("[bayes_skopt_amd] ensemble sampler: this run is driven from the host, one device batch per half-step (%s)" % reason,
              file=sys.stderr, flush=True)


def walkers_independent(coords):
    """True when the ensemble spans the parameter space: finite, non-degenerate columns, and a
    condition number of the centred/normalised ensemble <= 1e8 (emcee's initial-state check)."""
    coords = np.asarray(coords, dtype=np.float64)
    if not np.all(np.isfinite(coords)):
        return False
    C = coords - coords.mean(axis=0)[None, :]
    colmax = np.abs(C).max(axis=0)
    if np.any(colmax == 0):
        return False
    C = C / colmax
    C = C / np.sqrt((C * C).sum(axis=0))
    return np.linalg.cond(C.astype(float)) <= 1e8


class State:
    """Unpacks like emcee's State: ``coords, log_prob, random_state = sampler.run_mcmc(...)``."""

    def __init__(self, coords, log_prob, random_state):
        self.coords, self.log_prob, self.random_state = coords, log_prob, random_state

    def __iter__(self):
        return iter((self.coords, self.log_prob, self.random_state))


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, log_prob_fn, args=None, kwargs=None, a=2.0, vectorize=True,
                 live_dangerously=False, threads=None, **_ignored):
        """log_prob_fn: callable mapping an (Ns, ndim) block to (Ns,) log-probabilities when
        ``vectorize`` (default), else a per-walker callable (mapped over rows)."""
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self.log_prob_fn = log_prob_fn
        self.args = tuple(args) if args is not None else ()
        self.kwargs = dict(kwargs) if kwargs is not None else {}
        self.a = float(a)
        self.vectorize = vectorize
        self.live_dangerously = live_dangerously
        self._random = np.random.RandomState()
        self.reset()

    # -- emcee-compatible RNG plumbing (bask/bayesgpr.py:518-521 assigns a get_state() tuple) --
    @property
    def random_state(self):
        return self._random.get_state()

    @random_state.setter
    def random_state(self, state):
        try:
            self._random.set_state(state)
        except Exception:
            pass

    def reset(self):
        self._chain = None
        self._log_prob = None
        self.iteration = 0
        self.naccepted = np.zeros(self.nwalkers, dtype=np.int64)
        self.n_log_prob_evals = 0

    def _check_coords(self, coords):
        p = np.asarray(coords, dtype=np.float64)
        if not np.all(np.isfinite(p)):
            if np.any(np.isinf(p)):
                raise ValueError("At least one parameter value was infinite")
            raise ValueError("At least one parameter value was NaN")
        return p

    def _check_log_prob(self, lp, n):
        lp = np.asarray(lp, dtype=np.float64)
        if lp.shape != (n,):
            raise ValueError(f"log_prob_fn returned shape {lp.shape}, expected ({n},)")
        if np.any(np.isnan(lp)):
            raise ValueError("Probability function returned NaN")
        self.n_log_prob_evals += n
        return lp

    def compute_log_prob(self, coords):
        p = self._check_coords(coords)
        if self.vectorize:
            lp = self.log_prob_fn(p, *self.args, **self.kwargs)
        else:
            lp = np.array([float(self.log_prob_fn(row, *self.args, **self.kwargs)) for row in p])
        return self._check_log_prob(lp, p.shape[0])

    def _half_step_plans(self, nsteps):
        """Everything of a half-step that comes from the generator alone, in emcee's stream order: per step the move
        selection and the red/blue shuffle, per half-step the stretch factors and the partner draw.  None of it depends
        on a log-probability, so run_mcmc pulls the next plan while the device still works on the current block."""
        rng, a = self._random, self.a
        all_inds = np.arange(self.nwalkers)
        for _ in range(nsteps):
            rng.random_sample()  # move selection among a single StretchMove: choice(1, p=[1.0]) draws exactly this one double
            inds = all_inds % 2
            rng.shuffle(inds)
            for split in (0, 1):
                movers = np.flatnonzero(inds == split)
                others = np.flatnonzero(inds != split)
                Ns, Nc = movers.shape[0], others.shape[0]
                zz = ((a - 1.0) * rng.rand(Ns) + 1.0) ** 2.0 / a
                factors = (self.ndim - 1.0) * np.log(zz)
                partners = others[rng.randint(Nc, size=(Ns,))]
                yield movers, partners, zz[:, None], factors

    def run_mcmc(self, initial_state, nsteps, progress=False, skip_initial_state_check=False, log_prob0=None):
        coords = np.array(initial_state, dtype=np.float64, copy=True)
        if coords.shape != (self.nwalkers, self.ndim):
            raise ValueError("incompatible input dimensions")
        if self.nwalkers < 2 * self.ndim and not self.live_dangerously:
            raise RuntimeError(
                "It is unadvisable to use a red-blue move with fewer walkers than twice the number of dimensions."
            )
        if not skip_initial_state_check and not walkers_independent(coords):
            raise ValueError(
                "Initial state has a large condition number. "
                "Make sure that your walkers are linearly independent for the best performance"
            )
        log_prob = self.compute_log_prob(coords) if log_prob0 is None else np.array(log_prob0, dtype=np.float64)
        if np.any(np.isnan(log_prob)):
            raise ValueError("The initial log_prob was NaN")

        nsteps = int(nsteps)
        # a log_prob_fn that can describe itself to the device (BayesGPR with its default prior families, warped or not, sharded
        # over an RCCL group or not) runs the whole loop there: proposals, priors, LML batches, accept tests and the chain stay in
        # HBM; a progress bar follows the device through the plan's segments.  A run that cannot says why on stderr, once.
        resident = getattr(self.log_prob_fn, "resident", None) if (self.vectorize and nsteps > 0) else None
        if resident is not None:
            run = resident(self.nwalkers, self.ndim, *self.args, **self.kwargs)
            if run is not None:
                return self._run_resident(run, coords, log_prob, nsteps, progress)
            _tell_once(getattr(self.log_prob_fn, "resident_reason", None))
        chain = np.empty((nsteps, self.nwalkers, self.ndim))
        lps = np.empty((nsteps, self.nwalkers))
        rng = self._random
        # a log_prob_fn with begin()/finish() takes the block asynchronously (BayesGPR: the device factorises while
        # the host goes on); any other callable is evaluated in place.  The generator is consumed in the same order
        # either way: proposal draws, accept draws, next half-step's draws.
        begin = getattr(self.log_prob_fn, "begin", None) if self.vectorize else None
        finish = getattr(self.log_prob_fn, "finish", None) if begin is not None else None
        plans = self._half_step_plans(nsteps)
        plan = next(plans, None)
        pbar = _progress(progress, nsteps)
        for step in range(nsteps):
            for _split in (0, 1):
                movers, partners, zz, factors = plan
                Ns = movers.shape[0]
                s = coords[movers]
                cr = coords[partners]
                q = cr - (cr - s) * zz
                if finish is not None:
                    token = begin(self._check_coords(q), *self.args, **self.kwargs)  # <- one batched device call
                else:
                    new_lp = self.compute_log_prob(q)
                try:
                    with np.errstate(divide="ignore"):
                        logu = np.log(rng.rand(Ns))  # same stream as Ns scalar draws
                    plan = next(plans, None)
                finally:
                    if finish is not None:
                        new_lp = self._check_log_prob(finish(token), Ns)
                acc = factors + new_lp - log_prob[movers] > logu
                idx = movers[acc]
                coords[idx] = q[acc]
                log_prob[idx] = new_lp[acc]
                self.naccepted[idx] += 1
            chain[step] = coords
            lps[step] = log_prob
            pbar.update(1)
        pbar.close()
        self._chain = chain if self._chain is None else np.concatenate([self._chain, chain])
        self._log_prob = lps if self._log_prob is None else np.concatenate([self._log_prob, lps])
        self.iteration += nsteps
        return State(coords, log_prob, rng.get_state())

    def _run_resident(self, run, coords, log_prob, nsteps, progress=False):
        """The same run with the state on the device (``bgp_mcmc_begin`` / ``_steps`` / ``_end``): every draw of the generator
        is made here, in the order ``run_mcmc``'s loop makes them -- a half-step's stretch factors and partners, its accept
        draws, the next half-step's -- and handed over as the plan in growing segments, the next one drawn while the device
        works through the last; the generator ends in the same state and the device replays the same moves.  ``progress``: the
        bar (emcee's tqdm bar, ``bask/bayesgpr.py:522-524``) advances by the segments the device has worked through
        (``bgp_mcmc_progress``, polled between the segments and until the last one has passed)."""
        rng, Ns = self._random, self.nwalkers // 2
        state0 = rng.get_state()
        plans = self._half_step_plans(nsteps)
        run.begin(coords, log_prob, nsteps)
        pbar = _progress(progress, nsteps)
        live = not isinstance(pbar, _NoBar) and getattr(run, "progress", None) is not None
        shown = 0
        try:
            done, seg = 0, 2
            while done < nsteps:
                seg = min(seg, nsteps - done)
                movers = np.empty((2 * seg, Ns), dtype=np.int32)
                partners = np.empty((2 * seg, Ns), dtype=np.int32)
                zz, factors, logu = np.empty((2 * seg, Ns)), np.empty((2 * seg, Ns)), np.empty((2 * seg, Ns))
                for h in range(2 * seg):
                    mv, pt, z, f = next(plans)
                    movers[h], partners[h], zz[h], factors[h] = mv, pt, z[:, 0], f
                    with np.errstate(divide="ignore"):
                        logu[h] = np.log(rng.rand(Ns))
                run.steps((movers, partners, zz, factors, logu))
                done += seg
                seg = min(2 * seg, 64)
                if live:
                    now = run.progress()
                    pbar.update(now - shown)
                    shown = now
            while live and shown < nsteps:  # every segment is on the device: follow it to the end of the plan
                now = run.progress()
                if now == shown:
                    time.sleep(2e-3)
                pbar.update(now - shown)
                shown = now
        except BaseException:
            pbar.close()
            run.abandon()
            raise
        try:
            chain, lps, coords, log_prob, nacc, info = run.end()
        finally:
            pbar.update(nsteps - shown if not live else 0)
            pbar.close()
        if info[0]:
            # the host-driven loop raises from ``_check_coords`` in the half-step that proposed the value, with the generator
            # advanced that far and no further: the same message, the same generator state
            hf = int(info[2]) if len(info) > 2 else -1
            if hf >= 0:
                rng.set_state(state0)
                replay = self._half_step_plans(nsteps)
                for h in range(hf + 1):
                    next(replay)
                    if h < hf:
                        rng.rand(Ns)
            if len(info) > 3 and info[3]:
                raise ValueError("At least one parameter value was NaN")
            raise ValueError("At least one parameter value was infinite")
        self.n_log_prob_evals += 2 * nsteps * Ns
        self.naccepted += nacc
        self._chain = chain if self._chain is None else np.concatenate([self._chain, chain])
        self._log_prob = lps if self._log_prob is None else np.concatenate([self._log_prob, lps])
        self.iteration += nsteps
        self.resident_runs = getattr(self, "resident_runs", 0) + 1
        return State(coords, log_prob, rng.get_state())

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.iteration, 1)

    def _slice(self, arr, flat, discard, thin):
        if arr is None:
            raise AttributeError("you must run the sampler before accessing the chain")
        v = arr[discard + thin - 1 : self.iteration : thin]
        if flat:
            v = v.reshape((-1,) + v.shape[2:])
        return v

    def get_chain(self, flat=False, discard=0, thin=1):
        """(steps, W, ndim), or step-major (steps*W, ndim) when flat -- emcee's layout."""
        return self._slice(self._chain, flat, int(discard), int(thin))

    def get_log_prob(self, flat=False, discard=0, thin=1):
        return self._slice(self._log_prob, flat, int(discard), int(thin))


class _NoBar:
    def update(self, n):
        pass

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _progress(progress, total):
    if progress:
        try:
            import tqdm

            return tqdm.tqdm(total=total)
        except Exception:
            pass
    return _NoBar()
