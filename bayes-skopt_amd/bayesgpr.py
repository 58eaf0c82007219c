"""BayesGPR: fully Bayesian Gaussian-process regressor whose posterior arithmetic runs on the MI355X.

Host-side mirror of ``bask.BayesGPR`` (``bask/bayesgpr.py``): same constructor, ``fit`` / ``sample`` /
``predict`` / ``sample_y`` / ``theta`` / ``noise_set_to_zero`` surface and the same derived
quantities (walker count, step count, start ball, chain flattening, geometric median, warm start
through ``pos_``).  Everything numerical -- kernel matrices, Cholesky factorisations, solves, the
log-marginal likelihood evaluated for every MCMC proposal, predictive means / variances -- is done
by the HIP kernels behind the C-ABI (``include/bgp.h``); there is no CPU fallback.

Where the reference relies on inherited third-party code, the behaviour restated here is:
  * skopt's GaussianProcessRegressor.fit: append ``+ WhiteKernel()`` when ``noise="gaussian"``,
    MAP-fit theta, remember ``noise_``, replace the fitted WhiteKernel by ``WhiteKernel(0.0)`` in
    ``kernel_`` (so ``theta[-1] == -inf`` afterwards), keep factors that include the noise
    (SURVEY.md 2b, Appendix B.1);
  * sklearn's fit: y normalisation, L-BFGS-B on -LML from ``kernel_.theta`` within ``kernel_.bounds``
    (``sklearn/_gpr.py:296-341``) -- objective and gradient are evaluated on the device;
  * skopt's predict: ``var = diag - einsum(K*, K*, K_inv_)`` clipped at 0.
"""
import warnings
from contextlib import contextmanager, nullcontext

import numpy as np
import scipy.optimize
from sklearn.base import BaseEstimator, RegressorMixin, clone
from sklearn.utils import check_random_state

from . import _lib, distributed
from .kernels import ConstantKernel, WhiteKernel, analyse_kernel, param_for_white_kernel_in_sum
from .kernels import RBF as _RBF
from .sampler import EnsembleSampler
from .utils import geometric_median, guess_priors, validate_zeroone

__all__ = ["BayesGPR"]

_PD_MESSAGE = (
    "The kernel, %s, is not returning a positive definite matrix. Try gradually increasing the "
    "'alpha' parameter of your GaussianProcessRegressor estimator."
)


class BayesGPR(RegressorMixin, BaseEstimator):
    """Gaussian process regressor of which the kernel hyper-parameters are inferred in a fully
    Bayesian framework (constructor arguments as ``bask/bayesgpr.py:148-159``).

    Extra, MI355X-specific keyword: ``device`` (HIP device ordinal, default 0),
    ``max_batch`` (matrices factorised concurrently; default = half the walkers) and ``mvn`` -- how
    ``sample_y`` turns a predictive mean / covariance into function draws:

    * ``"reference"``: mean and covariance come from the device, the draw is numpy's legacy
      ``RandomState.multivariate_normal`` (SVD) on the host, exactly what the reference does
      (``sklearn/_gpr.py:522-526`` behind ``bask/bayesgpr.py:669-718``): a seeded call returns the
      reference's own variates;
    * ``"cholesky"``: ``mean + chol(cov + jitter I) z`` entirely on the device (same distribution, other
      variates; the only practical choice for thousands of query points, where the SVD takes minutes);
    * ``"auto"`` (default): ``"reference"`` up to ``MVN_REFERENCE_MAX_POINTS`` (512) query points,
      ``"cholesky"`` beyond.  The generator is consumed identically in both modes.

    ``resident_sampler`` (default True): ``sample`` / ``fit`` run the ensemble sampler with its walkers, proposals,
    log-priors, accept tests and chain resident on the device (``bgp_mcmc_begin_ex`` / ``_steps`` / ``_end``: no transfer
    between the first and the last half-step) whenever the kernel has a canonical device form and every prior is one of
    ``guess_priors``' two families or a frozen ``scipy.stats.norm`` (the default warp priors) -- with or without a progress
    bar, warped inputs, or an ensemble sharded over an RCCL group; the same moves as the host-driven loop, log-probabilities
    equal to ~1e-15 relative (the device's exp in the priors instead of numpy's).  A run that has to be driven from the
    host (custom priors, an odd ensemble, a generic kernel tree, a gloo group) says so once on stderr.
    """

    MVN_REFERENCE_MAX_POINTS = 512

    def __init__(
        self,
        kernel=None,
        alpha=1e-10,
        optimizer="fmin_l_bfgs_b",
        n_restarts_optimizer=0,
        normalize_y=False,
        warp_inputs=False,
        copy_X_train=True,
        random_state=None,
        noise="gaussian",
        device=0,
        max_batch=None,
        shard_ensemble=False,
        mvn="auto",
        resident_sampler=True,
    ):
        self._kernel = None if kernel is None else kernel.clone_with_theta(kernel.theta)
        self.kernel = kernel
        self.alpha = alpha
        self._alpha = alpha
        self.optimizer = optimizer
        self.n_restarts_optimizer = n_restarts_optimizer
        self.normalize_y = normalize_y
        self.warp_inputs = bool(warp_inputs)
        self.copy_X_train = copy_X_train
        self.random_state = check_random_state(random_state)
        self.noise = noise
        self.device = device
        self.max_batch = max_batch
        # exact single-ensemble sharding over the ranks of an initialised process group
        # (distributed.shard_log_prob; every rank must be constructed with the same random_state/data)
        self.shard_ensemble = bool(shard_ensemble)
        if mvn not in ("auto", "reference", "cholesky"):
            raise ValueError("mvn must be 'auto', 'reference' or 'cholesky', got %r" % (mvn,))
        self.mvn = mvn
        # the ensemble sampler's whole run on the device (bgp_mcmc_run) where the priors and the kernel allow it; False: the
        # host-driven loop (one LML batch per half-step), whose priors are numpy's to the last bit
        self.resident_sampler = bool(resident_sampler)
        self._sampler = None
        self.chain_ = None
        self.pos_ = None
        self.kernel_ = None
        self.noise_ = None
        self._ctx_obj = None  # device context (ctypes handle; never pickled, rebuilt on demand)
        self._needs_rebuild = False
        self._plan = None
        self._post_theta = None  # theta the resident posterior (L_, alpha_, K_inv_) was built with
        self._L = self._K_inv = None
        self.alpha_ = None

    # ------------------------------------------------------------------ device plumbing
    @property
    def _ctx(self):
        """The device context.  After unpickling / deep-copying (the ctypes handle does not travel) it is
        rebuilt from the stored training data the first time something needs it."""
        if self._ctx_obj is None and self._needs_rebuild:
            self._needs_rebuild = False
            saved = (self._post_theta, self.alpha_, self._L, self._K_inv)
            self._ensure_context()
            self._post_theta, self.alpha_, self._L, self._K_inv = saved
        return self._ctx_obj

    @_ctx.setter
    def _ctx(self, value):
        self._ctx_obj = value

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_ctx_obj"] = None
        state["_sampler"] = None  # holds the bound (possibly rank-sharded) log-probability
        state["_needs_rebuild"] = getattr(self, "_X_train_", None) is not None and self.kernel_ is not None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    def _ensure_context(self, batch_hint=None):
        """(Re)create / update the device context for the current training set."""
        X, y = self._X_train_, self.y_train_
        n = X.shape[0]
        alpha_diag = np.broadcast_to(np.asarray(self.alpha, dtype=np.float64), (n,)) if not np.iterable(self.alpha) \
            else np.asarray(self.alpha, dtype=np.float64)
        if alpha_diag.shape[0] != n:
            raise ValueError(f"alpha must be a scalar or an array with same number of entries as y. "
                             f"({alpha_diag.shape[0]} != {n})")
        plan = analyse_kernel(self.kernel_)
        want_batch = int(self.max_batch or batch_hint or getattr(self, "_batch_wish", None) or 64)
        ctx = self._ctx_obj
        if (ctx is None or ctx.d != X.shape[1] or ctx.form != plan.form or ctx.stationary != plan.stationary
                or ctx.max_batch < want_batch):
            if ctx is not None:
                ctx.close()
            ctx = _lib.Context(X, y, alpha_diag, form=plan.form, stationary=plan.stationary,
                               max_batch=max(want_batch, ctx.max_batch if ctx is not None else 0), device=self.device)
        else:
            ctx.update_data(X, y, alpha_diag)
        self._ctx, self._plan = ctx, plan
        self._post_theta = None
        self._L = self._K_inv = None
        self._install_warp()
        return ctx

    def _warp_vector(self):
        """[wa_1..wa_d, wb_1..wb_d] of the current warpers, or None (identity)."""
        if self.warp_inputs and hasattr(self, "warp_alphas_"):
            return np.concatenate([self.warp_alphas_, self.warp_betas_])
        return None

    def _install_warp(self):
        """Make the device context see the training inputs (and later query points) through the current
        warpers -- the role of ``rewarp()`` in the reference (``bask/bayesgpr.py:284-295``)."""
        if self._ctx is not None:
            self._ctx.set_warp(self._warp_vector())
            self._post_theta = None
            self._L = self._K_inv = None

    def _canonical(self, theta):
        return self._plan.canonical(theta, self._X_train_.shape[1])

    # ---- generic kernel expression trees (kernels.GramPlan): the host evaluates the kernel object, as the reference does for
    # every kernel (sklearn/_gpr.py:582); the device factorises, solves, inverts and forms the predictive products
    @property
    def _generic(self):
        return self._plan is not None and self._plan.generic

    def _kernel_at(self, theta):
        """The kernel object at ``theta`` (``kernel_.clone_with_theta``, sklearn/_gpr.py:571-576); log(0) noise levels pass."""
        with np.errstate(divide="ignore"):
            return self.kernel_.clone_with_theta(np.asarray(theta, dtype=np.float64))

    def _gram_stack(self, Thetas, X=None):
        """(B, n, n) stack of ``kernel(X_train)`` for every row of ``Thetas`` (no alpha: the device adds it)."""
        X = self.X_train_ if X is None else X
        return np.stack([self._kernel_at(t)(X) for t in np.atleast_2d(Thetas)])

    def _gram_lml(self, Thetas, Ws=None):
        """LML of every row (host kernel matrices, device factorisation).  ``Ws``: per-row input warps (B, 2d)."""
        Thetas = np.atleast_2d(Thetas)
        if Ws is None:
            K = self._gram_stack(Thetas)
        else:  # every walker sees the training inputs through its own Beta-CDF warp (bask/bayesgpr.py:353-365)
            K = np.stack([self._kernel_at(t)(self._ctx.beta_cdf(self._X_train_, w)) for t, w in zip(Thetas, Ws)])
        return self._ctx.lml_gram(K)

    def log_marginal_likelihood(self, theta=None, eval_gradient=False, clone_kernel=True):
        """``sklearn/_gpr.py:537-652`` on the device.  theta may be (p,) or a (B, p) block."""
        if theta is None:
            if eval_gradient:
                raise ValueError("Gradient can only be evaluated for theta!=None")
            return self.log_marginal_likelihood_value_
        theta = np.asarray(theta, dtype=np.float64)
        single = theta.ndim == 1
        if self._generic:
            return self._gram_lml_and_grad(theta, eval_gradient)
        H = self._canonical(theta)
        if eval_gradient:
            lml, gh, _ = self._ctx.lml_grad(H)
            g = self._plan.grad_to_theta(gh, self._X_train_.shape[1])
            return (float(lml[0]), g[0]) if single else (lml, g)
        lml = self._ctx.lml(H)
        return float(lml[0]) if single else lml

    def _gram_lml_and_grad(self, theta, eval_gradient):
        """``sklearn/_gpr.py:579-647`` for a generic tree: ``K, K_gradient = kernel(X, eval_gradient=True)`` on the host;
        factorisation, alpha, K^-1 and the LML on the device; the contraction ``1/2 sum_ij (alpha_i alpha_j - K^-1_ij)
        dK_ij/dtheta_k`` of the host-evaluated gradient tensor with them (``:615-647``)."""
        single = theta.ndim == 1
        T = np.atleast_2d(theta)
        if not eval_gradient:
            lml = self._gram_lml(T)
            return float(lml[0]) if single else lml
        X = self.X_train_
        vals, grads = np.empty(len(T)), np.empty((len(T), T.shape[1]))
        self._gram_resident = None  # (posterior_gram below overwrites the device-resident K^-1 / alpha of theta)
        for i, t in enumerate(T):
            K, Kg = self._kernel_at(t)(X, eval_gradient=True)
            res = self._ctx.posterior_gram(K, want_alpha=True, want_K_inv=True)
            if res["status"][0] != 0:
                vals[i], grads[i] = -np.inf, 0.0
                continue
            a, Ki = res["alpha"][0], res["K_inv"][0]
            vals[i] = res["lml"][0]
            grads[i] = 0.5 * (np.einsum("i,ijk,j->k", a, Kg, a) - np.einsum("ij,ijk->k", Ki, Kg))
        return (float(vals[0]), grads[0]) if single else (vals, grads)

    # ------------------------------------------------------------------ theta / posterior
    @property
    def theta(self):
        """Current (geometric-median) kernel hyper-parameters in log space
        (``bask/bayesgpr.py:182-198``)."""
        if self.kernel_ is not None:
            with np.errstate(divide="ignore"):
                return np.copy(self.kernel_.theta)
        return None

    @theta.setter
    def theta(self, theta):
        """Posterior build, ``bask/bayesgpr.py:200-217``: K, L_, K_inv_, alpha_ on the device."""
        theta = np.asarray(theta, dtype=np.float64)
        self.kernel_.theta = theta
        self._build_posterior(theta)

    def _build_posterior(self, theta):
        if self._generic:
            res = self._ctx.posterior_gram(self._gram_stack(theta), want_alpha=True)
        else:
            res = self._ctx.posterior(self._canonical(theta), want_L=False, want_alpha=True, want_K_inv=False)
        if res["status"][0] != 0:
            raise np.linalg.LinAlgError(
                _PD_MESSAGE % self.kernel_,
                "%d-th leading minor of the array is not positive definite" % res["status"][0],
            )
        self.alpha_ = res["alpha"][0]
        self._post_theta = np.array(theta, copy=True)
        self._gram_resident = np.array(theta, copy=True) if self._generic else None
        self._L = self._K_inv = None

    def _fetch_factor(self, which):
        if self._post_theta is None:
            raise AttributeError("no posterior has been built yet")
        if self._generic:
            res = self._ctx.posterior_gram(self._gram_stack(self._post_theta), want_L=(which == "L"), want_alpha=False,
                                           want_K_inv=(which == "K_inv"))
            self._gram_resident = np.array(self._post_theta, copy=True)
            return res[which][0]
        res = self._ctx.posterior(self._canonical(self._post_theta), want_L=(which == "L"), want_alpha=False,
                                  want_K_inv=(which == "K_inv"))
        return res[which][0]

    @property
    def L_(self):
        """Lower Cholesky factor of the kernel matrix (fetched from the device on demand)."""
        if self._L is None:
            self._L = self._fetch_factor("L")
        return self._L

    @L_.setter
    def L_(self, value):
        self._L = value

    @property
    def K_inv_(self):
        """Explicit inverse K^-1 (``bask/bayesgpr.py:207-208``), fetched on demand."""
        if self._K_inv is None:
            self._K_inv = self._fetch_factor("K_inv")
        return self._K_inv

    @K_inv_.setter
    def K_inv_(self, value):
        self._K_inv = value

    @property
    def X_train_(self):
        """Training inputs; the WARPED ones when ``warp_inputs`` and warpers exist
        (``bask/bayesgpr.py:219-235``)."""
        X = getattr(self, "_X_train_", None)
        if X is not None and self.warp_inputs and hasattr(self, "warp_alphas_") and self._ctx is not None:
            return self.warp(X)
        return X

    @X_train_.setter
    def X_train_(self, X_train):
        X_train = np.asarray(X_train, dtype=np.float64)
        self._X_train_ = np.copy(X_train) if self.copy_X_train else X_train

    # ------------------------------------------------------------------ input warping
    def warp(self, X):
        """Beta-CDF warp of X with the current warpers, evaluated on the device; identity when
        ``warp_inputs=False`` or before the first fit (``bask/bayesgpr.py:249-264``)."""
        w = self._warp_vector()
        if w is None or self._ctx is None:
            return X
        return self._ctx.beta_cdf(np.atleast_2d(np.asarray(X, dtype=np.float64)), w)

    def unwarp(self, X):
        """Inverse warp (Beta quantile function per column, ``bask/bayesgpr.py:266-282``).  Only used to
        GENERATE candidate points (``bask/optimizer.py:353-357``); host-side scipy."""
        if not (self.warp_inputs and hasattr(self, "warp_alphas_")):
            return X
        import scipy.stats as st

        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        out = np.empty_like(X)
        for col, (a_log, b_log) in enumerate(zip(self.warp_alphas_, self.warp_betas_)):
            out[:, col] = st.beta(a=np.exp(a_log), b=np.exp(b_log)).ppf(X[:, col])
        return out

    def rewarp(self):
        """Apply the current warpers to the resident training inputs again."""
        if self.warp_inputs:
            self._install_warp()

    def create_warpers(self, alphas, betas):
        """Set the Beta-CDF parameters (log space) of every input column
        (``bask/bayesgpr.py:297-316``).  ``warpers_`` / ``unwarpers_`` are kept as host callables for
        API compatibility; the device evaluates the warp itself."""
        if self.warp_inputs:
            import scipy.stats as st

            self.warp_alphas_ = np.copy(np.asarray(alphas, dtype=np.float64))
            self.warp_betas_ = np.copy(np.asarray(betas, dtype=np.float64))
            dists = [st.beta(a=np.exp(a), b=np.exp(b)) for a, b in zip(self.warp_alphas_, self.warp_betas_)]
            self.warpers_ = [dist.cdf for dist in dists]
            self.unwarpers_ = [dist.ppf for dist in dists]

    @contextmanager
    def noise_set_to_zero(self):
        """Context in which kernel_'s white-noise level is 0 WITHOUT recomputing alpha_ / L_ /
        K_inv_ (``bask/bayesgpr.py:318-336``)."""
        current_theta = self.theta
        try:
            present, white_param = param_for_white_kernel_in_sum(self.kernel_)
            self.kernel_.set_params(**{white_param: WhiteKernel(noise_level=0.0)})
            yield self
        finally:
            self.kernel_.theta = current_theta

    def _apply_noise_vector(self, n_instances, noise_vector):
        """``bask/bayesgpr.py:338-349``."""
        if noise_vector is not None:
            base = self.alpha if not np.iterable(self.alpha) else self._alpha
            if np.iterable(base):
                raise ValueError("alpha passed to the constructor must be a scalar when a noise_vector is used")
            alpha = np.ones(n_instances) * base
            alpha[: len(noise_vector)] += noise_vector
            self.alpha = alpha

    # ------------------------------------------------------------------ log posterior
    def _log_prob_batch(self, Theta, priors, warp_priors=None):
        """Vectorised ``_log_prob_fn`` (``bask/bayesgpr.py:351-379``): sum of priors (host) + LML
        (device) for a (Ns, p) block; non-finite -> -inf.  With ``warp_inputs`` the last 2d columns are
        the Beta-CDF parameters of each walker's own input warp (``:353-365``)."""
        return self._log_prob_finish(self._log_prob_begin(Theta, priors, warp_priors))

    def _log_prob_begin(self, Theta, priors, warp_priors=None):
        """First half of ``_log_prob_batch``: put the block's LML batch on the device and return at once with the
        priors of the same block (evaluated while the device factorises)."""
        Theta = np.atleast_2d(Theta)
        if self._generic:  # host kernel matrices: nothing to overlap the priors with
            d = self._X_train_.shape[1] if self.warp_inputs else 0
            Tgp, W = (Theta[:, : Theta.shape[1] - 2 * d], Theta[:, Theta.shape[1] - 2 * d :]) if d else (Theta, None)
            lp = _eval_priors(priors, Tgp)
            if W is not None:
                lp = lp + _eval_warp_priors(warp_priors, W, d)
            return lp, ("gram", Tgp, W), False
        if self.warp_inputs:
            d = self._X_train_.shape[1]
            Tgp, W = Theta[:, : Theta.shape[1] - 2 * d], Theta[:, Theta.shape[1] - 2 * d :]
            H = (self._canonical(Tgp), np.ascontiguousarray(W))
            submitted = self._ctx.lml_warped_submit(*H)
        else:
            Tgp, W, d = Theta, None, 0
            H = self._canonical(Theta)
            submitted = self._ctx.lml_submit(H)
        try:
            lp = _eval_priors(priors, Tgp)
            if W is not None:
                lp = lp + _eval_warp_priors(warp_priors, W, d)
        except BaseException:
            if submitted:
                self._ctx.lml_wait()
            raise
        return lp, H, submitted

    def _log_prob_finish(self, token):
        lp, H, submitted = token
        if submitted:
            lml = self._ctx.lml_wait()
        elif isinstance(H, tuple) and len(H) == 3 and isinstance(H[0], str):
            lml = self._gram_lml(H[1], H[2])
        elif self.warp_inputs:
            lml = self._ctx.lml_warped(*H)
        else:
            lml = self._ctx.lml(H)
        with np.errstate(invalid="ignore"):
            lp = lp + lml
        lp[~np.isfinite(lp)] = -np.inf
        return lp

    def _log_prob_fn(self, x, priors, warp_priors=None):
        return float(self._log_prob_batch(np.asarray(x, dtype=np.float64)[None, :], priors, warp_priors)[0])

    # ------------------------------------------------------------------ sample
    def sample(
        self,
        X=None,
        y=None,
        noise_vector=None,
        n_threads=1,
        n_desired_samples=100,
        n_burnin=0,
        n_thin=1,
        n_walkers_per_thread=100,
        progress=False,
        priors=None,
        warp_priors=None,
        position=None,
        add=False,
        **kwargs,
    ):
        """Sample the hyper-parameter posterior with the ensemble sampler
        (``bask/bayesgpr.py:381-548``; same arguments and derived quantities)."""
        if (X is None and self.X_train_ is None) or self.kernel_ is None:
            raise ValueError(
                "It looks like you are trying to sample from the GP posterior without data. "
                "Pass X and y, or ensure that you call fit before sample."
            )
        if priors is None:
            priors = guess_priors(self.kernel_)
        if warp_priors is None:
            import scipy.stats as st

            warp_priors = (st.norm(loc=0.0, scale=0.3).logpdf, st.norm(loc=0.0, scale=0.3).logpdf)

        if X is not None:
            X = np.asarray(X, dtype=np.float64)
            y = np.asarray(y, dtype=np.float64)
            if self.normalize_y:
                self._y_train_mean = np.mean(y, axis=0)
                self._y_train_std = np.std(y, axis=0)
            else:
                self._y_train_mean = np.zeros(1)
                self._y_train_std = 1
            self.y_train_std_ = self._y_train_std
            self.y_train_mean_ = self._y_train_mean
            y = (y - self.y_train_mean_) / self.y_train_std_
            if noise_vector is not None:
                noise_vector = np.array(noise_vector) / np.power(self.y_train_std_, 2)
            self.X_train_ = X
            self.y_train_ = np.copy(y) if self.copy_X_train else y

        self._apply_noise_vector(len(self.y_train_), noise_vector)

        n_dim = len(self.theta)
        n_walkers = n_threads * n_walkers_per_thread
        n_samples = int(np.ceil(n_desired_samples / n_walkers) + n_burnin)
        pos = None
        if position is not None:
            pos = position
        elif self.pos_ is not None:
            pos = self.pos_
        n_theta = n_dim
        if self.warp_inputs:
            added_dims = self._X_train_.shape[1] * 2
            n_dim += added_dims
        if pos is None:
            theta = self.theta
            theta[np.isinf(theta)] = np.log(self.noise_)
            if self.warp_inputs:
                theta = np.concatenate([theta, np.zeros(added_dims)])
            pos = [theta + 1e-2 * self.random_state.randn(n_dim) for _ in range(n_walkers)]

        if self.shard_ensemble:  # every rank must propose the same blocks: pin the start ensemble to rank 0's
            pos = distributed.broadcast_array(np.asarray(pos, dtype=np.float64))
        self._ensure_context(batch_hint=(n_walkers + 1) // 2)
        self._sampler = EnsembleSampler(
            nwalkers=n_walkers,
            ndim=n_dim,
            log_prob_fn=_ShardedLogProb(self) if self.shard_ensemble else _AsyncLogProb(self),
            kwargs=dict(priors=priors, warp_priors=warp_priors),
            threads=n_threads,
            **kwargs,
        )
        rng = np.random.RandomState(self.random_state.randint(0, np.iinfo(np.int32).max))
        self._sampler.random_state = rng.get_state()
        pos, prob, state = self._sampler.run_mcmc(pos, n_samples, progress=progress)
        chain = self._sampler.get_chain(flat=True, discard=n_burnin, thin=n_thin)
        if add and self.chain_ is not None:
            self.chain_ = np.concatenate([self.chain_, chain])
        else:
            self.chain_ = chain
        if self.warp_inputs:
            median = geometric_median(self.chain_)
            d = self._X_train_.shape[1]
            warp_params = median[n_theta:]
            self.create_warpers(warp_params[:d], warp_params[d:])
            self.rewarp()
            self.theta = median[:n_theta]
        else:
            self.theta = geometric_median(self.chain_)
        self.log_marginal_likelihood_value_ = self.log_marginal_likelihood(self.kernel_.theta, clone_kernel=False)
        self.pos_ = pos

    # ------------------------------------------------------------------ fit
    def fit(
        self,
        X,
        y,
        noise_vector=None,
        n_threads=1,
        n_desired_samples=100,
        n_burnin=10,
        n_walkers_per_thread=100,
        progress=True,
        priors=None,
        warp_priors=None,
        position=None,
        **kwargs,
    ):
        """MAP initialisation (L-BFGS-B on the device LML) followed by ``sample``
        (``bask/bayesgpr.py:550-620``)."""
        self.kernel = self._kernel
        X = np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        if self.normalize_y and noise_vector is not None:
            y_std = np.std(y, axis=0)
            noise_vector = np.array(noise_vector) / np.power(y_std, 2)
        self._apply_noise_vector(len(y), noise_vector)
        # the context the MAP start creates is the one the sampler wants (a second allocation of the matrix workspace otherwise:
        # 2 x 15 ms + 5 ms of release at n = 2048 x 128 matrices, tools/fit_cprofile.py)
        self._batch_wish = max(64, (int(n_threads) * int(n_walkers_per_thread) + 1) // 2)  # (64: the default without a wish)
        self._map_fit(X, y)
        self.sample(
            n_threads=n_threads,
            n_desired_samples=n_desired_samples,
            n_burnin=n_burnin,
            n_walkers_per_thread=n_walkers_per_thread,
            progress=progress,
            priors=priors,
            warp_priors=warp_priors,
            position=position,
            add=False,
            **kwargs,
        )
        return self

    def _map_fit(self, X, y):
        """skopt GPR.fit -> sklearn GPR.fit restated (module docstring) with device LML/gradient."""
        if isinstance(self.noise, str) and self.noise != "gaussian":
            raise ValueError("expected noise to be 'gaussian', got %s" % self.noise)
        if self.kernel is None:
            self.kernel = ConstantKernel(1.0, constant_value_bounds="fixed") * _RBF(1.0, length_scale_bounds="fixed")
        if self.noise and not param_for_white_kernel_in_sum(self.kernel)[0]:
            if self.noise == "gaussian":
                self.kernel = self.kernel + WhiteKernel()
            else:
                self.kernel = self.kernel + WhiteKernel(noise_level=self.noise, noise_level_bounds="fixed")
        self.kernel_ = clone(self.kernel)

        if self.normalize_y:
            self._y_train_mean = np.mean(y, axis=0)
            std = np.std(y, axis=0)
            self._y_train_std = 1.0 if std == 0.0 else std  # _handle_zeros_in_scale
            y = (y - self._y_train_mean) / self._y_train_std
        else:
            self._y_train_mean = np.zeros(1)
            self._y_train_std = np.ones(1)
        if np.iterable(self.alpha) and np.shape(self.alpha)[0] != y.shape[0]:
            if np.shape(self.alpha)[0] == 1:
                self.alpha = self.alpha[0]
            else:
                raise ValueError(
                    "alpha must be a scalar or an array with same number of "
                    f"entries as y. ({np.shape(self.alpha)[0]} != {y.shape[0]})"
                )
        self.X_train_ = X
        self.y_train_ = np.copy(y) if self.copy_X_train else y
        self.y_train_std_ = self._y_train_std
        self.y_train_mean_ = self._y_train_mean
        self._ensure_context()

        if self.optimizer is not None and self.kernel_.n_dims > 0:
            def obj(theta):
                lml, grad = self.log_marginal_likelihood(theta, eval_gradient=True, clone_kernel=False)
                return -lml, -grad

            optima = [self._constrained_optimization(obj, self.kernel_.theta, self.kernel_.bounds)]
            if self.n_restarts_optimizer > 0:
                bounds = self.kernel_.bounds
                if not np.isfinite(bounds).all():
                    raise ValueError(
                        "Multiple optimizer restarts (n_restarts_optimizer>0) requires that all bounds are finite."
                    )
                for _ in range(self.n_restarts_optimizer):
                    theta_initial = self.random_state.uniform(bounds[:, 0], bounds[:, 1])
                    optima.append(self._constrained_optimization(obj, theta_initial, bounds))
            vals = [o[1] for o in optima]
            self.kernel_.theta = optima[int(np.argmin(vals))][0]
            self.log_marginal_likelihood_value_ = -np.min(vals)
        else:
            self.log_marginal_likelihood_value_ = self.log_marginal_likelihood(self.kernel_.theta)

        # factors built WITH the fitted noise ...
        self._build_posterior(self.kernel_.theta)
        # ... then kernel_'s WhiteKernel is zeroed (skopt's post-fit step)
        self.noise_ = None
        if self.noise:
            if isinstance(self.kernel_, WhiteKernel):
                self.kernel_.set_params(noise_level=0.0)
            else:
                present, white_param = param_for_white_kernel_in_sum(self.kernel_)
                if present:
                    self.noise_ = self.kernel_.get_params()[white_param].noise_level
                    self.kernel_.set_params(**{white_param: WhiteKernel(noise_level=0.0)})

    def _constrained_optimization(self, obj_func, initial_theta, bounds):
        if self.optimizer == "fmin_l_bfgs_b":
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = scipy.optimize.minimize(obj_func, initial_theta, method="L-BFGS-B", jac=True, bounds=bounds)
            return res.x, res.fun
        if callable(self.optimizer):
            return self.optimizer(obj_func, initial_theta, bounds=bounds)
        raise ValueError(f"Unknown optimizer {self.optimizer}.")

    # ------------------------------------------------------------------ predict
    def _kernel_theta_for_predict(self):
        with np.errstate(divide="ignore"):
            return np.copy(self.kernel_.theta)

    def predict(self, X, return_std=False, return_cov=False, return_mean_grad=False, return_std_grad=False):
        """Predictive mean [, std | cov] (``bask/bayesgpr.py:622-635`` -> skopt predict)."""
        if return_std and return_cov:
            raise RuntimeError("Not returning standard deviation of predictions when returning full covariance.")
        if return_std_grad and not return_std:
            raise ValueError("Not returning std_gradient without returning the std.")
        if return_std_grad and not return_mean_grad:
            raise ValueError("Not returning std_gradient without returning the mean_gradient.")
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        if return_mean_grad and X.shape[0] > 1:
            raise NotImplementedError("Not implemented for n_samples > 1")
        if self.warp_inputs:
            validate_zeroone(X)  # the device warps the query points with the context-level warp
        if self._post_theta is None or getattr(self, "_X_train_", None) is None:
            raise RuntimeError("predict before fit is not supported on the MI355X path")
        if self._generic:
            mean, var, cov = self._rows_predict(None, X, noise_zero=False, return_cov=return_cov)
            y_mean = self.y_train_std_ * mean[0] + self.y_train_mean_
            if return_cov:
                return y_mean, cov[0] * self.y_train_std_**2
            y_std = np.sqrt(var[0] * self.y_train_std_**2)
            if return_mean_grad:
                grad_mean, grad_std = self._predict_gradients_generic(X[0], y_std, return_std_grad)
                if return_std_grad:
                    return y_mean, y_std, grad_mean, grad_std
                return (y_mean, y_std, grad_mean) if return_std else (y_mean, grad_mean)
            return (y_mean, y_std) if return_std else y_mean
        self._make_resident()
        Hk = self._canonical(self._kernel_theta_for_predict())
        if return_cov:
            mean, var, cov = self._ctx.predict(Hk, X, return_cov=True)
            y_mean = self.y_train_std_ * mean[0] + self.y_train_mean_
            return y_mean, cov[0] * self.y_train_std_**2
        mean, var = self._ctx.predict(Hk, X)
        y_mean = self.y_train_std_ * mean[0] + self.y_train_mean_
        y_std = np.sqrt(var[0] * self.y_train_std_**2)
        if return_mean_grad:
            grad_mean, grad_std = self._predict_gradients(X[0], Hk[0], y_std, return_std_grad)
            if return_std_grad:
                return y_mean, y_std, grad_mean, grad_std
            return (y_mean, y_std, grad_mean) if return_std else (y_mean, grad_mean)
        if return_std:
            return y_mean, y_std
        return y_mean

    def _predict_gradients(self, x, hk, y_std, want_std_grad):
        """Gradients of the predictive mean / std at ONE query point (skopt's
        ``GaussianProcessRegressor.predict(return_mean_grad, return_std_grad)``, the routine
        ``bask/bayesgpr.py:633`` forwards to): ``grad = kernel_.gradient_x(x, X_train_)`` (n, d),
        ``grad_mean = grad^T alpha_``, ``grad_std = -K_* K_inv_ grad / std``.  O(n d) + one O(n^2) product on
        the host from the device-built ``alpha_`` / ``K_inv_``; with input warping the derivative is with
        respect to the warped coordinates, as in the reference."""
        Xt = self.X_train_
        if self.warp_inputs:
            x = self.warp(x[None, :])[0]
        d = Xt.shape[1]
        ell2 = np.exp(2.0 * hk[1 : d + 1])
        diff = x[None, :] - Xt                       # (n, d)
        r = np.sqrt(np.sum(diff * diff / ell2, axis=1))
        stat = self._plan.stationary
        with np.errstate(divide="ignore", invalid="ignore"):
            if stat == "rbf":
                S = np.exp(-0.5 * r * r)
                fac = -S
            elif stat == "matern12":
                S = np.exp(-r)
                fac = np.where(r > 0, -S / r, 0.0)
            elif stat == "matern32":
                e = np.exp(-np.sqrt(3.0) * r)
                S = (1.0 + np.sqrt(3.0) * r) * e
                fac = -3.0 * e
            else:
                e = np.exp(-np.sqrt(5.0) * r)
                S = (1.0 + np.sqrt(5.0) * r + 5.0 / 3.0 * r * r) * e
                fac = -(5.0 / 3.0) * (1.0 + np.sqrt(5.0) * r) * e
        cst = np.exp(hk[0])
        scale = cst if self._plan.form == "product" else 1.0
        grad = scale * fac[:, None] * diff / ell2     # d k(x, X_i) / d x   (Constant / White terms: zero)
        grad_mean = (grad.T @ self.alpha_) * self.y_train_std_
        if not want_std_grad:
            return grad_mean, None
        grad_std = np.zeros(d)
        if not np.allclose(y_std, 0.0):
            k_trans = cst * S if self._plan.form == "product" else cst + S
            grad_std = -(k_trans @ (self.K_inv_ @ grad)) / y_std[0] * self.y_train_std_**2
        return grad_mean, grad_std

    def _predict_gradients_generic(self, x, y_std, want_std_grad):
        """The same two gradients for a generic kernel tree: ``kernel_.gradient_x`` is ``kernels.gradient_x`` (skopt's method
        restated for scikit-learn kernel objects), ``K_*`` comes from the host-evaluated kernel object, ``alpha_`` / ``K_inv_``
        from the device (``bgp_posterior_batch_gram``)."""
        from .kernels import gradient_x

        Xt = self.X_train_
        if self.warp_inputs:
            x = self.warp(x[None, :])[0]
        grad = gradient_x(self.kernel_, x, Xt)
        grad_mean = (grad.T @ self.alpha_) * self.y_train_std_
        if not want_std_grad:
            return grad_mean, None
        grad_std = np.zeros(Xt.shape[1])
        if not np.allclose(y_std, 0.0):
            k_trans = self.kernel_(x[None, :], Xt)[0]
            grad_std = -(k_trans @ (self.K_inv_ @ grad)) / y_std[0] * self.y_train_std_**2
        return grad_mean, grad_std

    def _make_resident(self):
        """Make sure the device holds the posterior that alpha_/L_/K_inv_ describe."""
        if self._generic:
            if getattr(self, "_gram_resident", None) is None or not np.array_equal(self._gram_resident, self._post_theta):
                self._ctx.posterior_gram(self._gram_stack(self._post_theta), want_alpha=False)
                self._gram_resident = np.array(self._post_theta, copy=True)
            return
        H = self._canonical(self._post_theta)
        res = self._ctx.resident_H
        if res is None or res.shape[0] < 1 or not np.array_equal(res[0], H[0]):
            self._ctx.posterior(H, want_alpha=False)

    def _predict_hyper_samples(self, thetas, X, noise_zero=True):
        """Posterior build + predict for a whole batch of hyper-posterior draws (what
        ``evaluate_acquisitions`` does one ``gpr.theta = chain_[i]`` at a time,
        ``bask/acquisition.py:112-125``): ONE batched device build, ONE batched predict."""
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        if self.warp_inputs:
            return self._predict_hyper_samples_warped(np.atleast_2d(thetas), X, noise_zero)
        if self._generic:
            mean, var, _ = self._rows_predict(np.atleast_2d(thetas), X, noise_zero)
            return self.y_train_std_ * mean + self.y_train_mean_, np.sqrt(var * self.y_train_std_**2)
        Hk = self._build_hyper_samples(thetas).copy()
        if noise_zero:
            Hk[:, -1] = -np.inf
        mean, var = self._ctx.predict(Hk, X)
        mu = self.y_train_std_ * mean + self.y_train_mean_
        return mu, np.sqrt(var * self.y_train_std_**2)

    def _rows_predict(self, thetas, X, noise_zero, return_cov=False):
        """Posterior build + predict for the GP of every row of ``thetas`` under the CURRENT warpers, in the units of the
        normalised targets: (mean, var, cov | None), each with a leading row axis.  ``thetas=None``: the resident posterior
        of ``theta`` with the kernel parameters currently in ``kernel_`` (what ``predict`` needs).  Canonical kernels: one
        batched device build over the distinct rows + one batched device predict.  Generic trees: the kernel object is
        evaluated on the host (training matrix, cross covariances, prior variances), everything else on the device."""
        if not self._generic:
            H = self._canonical(thetas)
            uniq, inverse = np.unique(H, axis=0, return_inverse=True)
            res = self._ctx.posterior(uniq, want_alpha=False)
            self._raise_if_not_pd(res["status"])
            Hk = uniq.copy()
            if noise_zero:
                Hk[:, -1] = -np.inf
            out = self._ctx.predict(Hk, X, return_cov=return_cov)
            inverse = np.asarray(inverse).ravel()
            return out[0][inverse], out[1][inverse], (out[2][inverse] if return_cov else None)
        Xw = self.warp(X) if self.warp_inputs else X  # BayesGPR.predict warps the query points (bask/bayesgpr.py:630-632)
        Xt = self.X_train_
        if thetas is None:
            self._make_resident()
            kernels = [self.kernel_]
        else:
            thetas = np.atleast_2d(thetas)
            self._gram_resident = None
            res = self._ctx.posterior_gram(self._gram_stack(thetas, Xt), want_alpha=False)
            self._raise_if_not_pd(res["status"])
            kernels = [self._kernel_at(t) for t in thetas]
        if noise_zero:
            kernels = [_with_white_zeroed(k) for k in kernels]
        Ks = np.stack([k(Xw, Xt) for k in kernels])
        kss = np.stack([k.diag(Xw) for k in kernels])
        Kss = np.stack([k(Xw) for k in kernels]) if return_cov else None
        out = self._ctx.predict_gram(Ks, kss, Kss)
        return out[0], out[1], (out[2] if return_cov else None)

    def _raise_if_not_pd(self, status):
        if np.any(status != 0):
            bad = int(np.flatnonzero(status)[0])
            raise np.linalg.LinAlgError(
                _PD_MESSAGE % self.kernel_,
                "%d-th leading minor of the array is not positive definite" % status[bad],
            )

    def _build_hyper_samples(self, thetas):
        """One batched device posterior build for a set of chain rows; returns the canonical hyper-parameters."""
        H = self._canonical(np.atleast_2d(thetas))
        res = self._ctx.posterior(H, want_alpha=False)
        if np.any(res["status"] != 0):
            b = int(np.flatnonzero(res["status"])[0])
            raise np.linalg.LinAlgError(
                _PD_MESSAGE % self.kernel_,
                "%d-th leading minor of the array is not positive definite" % res["status"][b],
            )
        return H

    def _acq_hyper_samples(self, thetas, X, kinds, params, n_samples, noise_zero=True):
        """Closed-form acquisition values averaged over a batch of hyper-posterior draws, entirely on the device:
        batched posterior build, batched predict and the acquisition / averaging pass of
        ``evaluate_acquisitions`` (``bask/acquisition.py:112-139``) -- only (len(kinds), m) numbers come back."""
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        Hk = self._build_hyper_samples(thetas).copy()
        if noise_zero:
            Hk[:, -1] = -np.inf
        y_mean, y_std = float(np.ravel(self.y_train_mean_)[0]), float(np.ravel(self.y_train_std_)[0])
        return self._ctx.acq(Hk, X, y_mean, y_std, kinds, params, n_samples)

    def _predict_hyper_samples_warped(self, rows, X, noise_zero):
        """With input warping every hyper-posterior draw carries its own warp, i.e. its own training
        inputs: one posterior build + predict per draw (``bask/acquisition.py:113-119``), all on the
        device; the warpers are restored afterwards (``:142-145``)."""
        validate_zeroone(X)
        n_theta = len(self.kernel_.theta)
        d = self._X_train_.shape[1]
        backup = (np.copy(self.warp_alphas_), np.copy(self.warp_betas_))
        mus, stds = [], []
        for row in rows:
            self.create_warpers(row[n_theta : n_theta + d], row[n_theta + d :])
            self.rewarp()
            mean, var, _ = self._rows_predict(row[None, :n_theta], X, noise_zero)
            mus.append(self.y_train_std_ * mean[0] + self.y_train_mean_)
            stds.append(np.sqrt(var[0] * self.y_train_std_**2))
        self.create_warpers(*backup)
        self.rewarp()
        return np.array(mus), np.array(stds)

    def _pvrs(self, X, thompson_points, has_alpha_vec):
        """Device side of PVRS / VarianceReduction (``bask/acquisition.py:287-300,328-338``)."""
        if self._generic:
            return self._pvrs_gram(np.atleast_2d(X), np.atleast_2d(thompson_points), has_alpha_vec)
        Hk = self._canonical(self._kernel_theta_for_predict())
        status = self._ctx.pvrs_prepare(Hk, has_alpha_vec)
        if status != 0:
            raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % status)
        return self._ctx.pvrs(Hk, X, np.atleast_2d(thompson_points))

    def _pvrs_gram(self, X, T, has_alpha_vec):
        """PVRS for a generic kernel tree through the bordered-inverse identity of ``bgp_pvrs``,
        ``covs_i = sum_t [k_t^T K^-1 k_t + (k(x_t, x_i) - k_i^T K^-1 k_t)^2 / (kappa_i - k_i^T K^-1 k_i)]``, every term read off
        ONE device predictive covariance ``C = K_** - K_* K^-1 K_*^T`` over [Thompson points; candidates] per candidate chunk:
        ``C_ti``, ``C_ii`` and ``k_t^T K^-1 k_t = kappa_t - C_tt``.  K carries alpha only when it is a vector (reference quirk,
        ``bask/acquisition.py:332-333``); the kernel matrices come from the host-evaluated ``kernel_``."""
        k = self.kernel_
        Xt = self.X_train_
        self._gram_resident = None
        res = self._ctx.posterior_gram(k(Xt)[None], use_alpha=bool(has_alpha_vec), want_alpha=False)
        self._raise_if_not_pd(res["status"])
        if self.warp_inputs:
            X, T = self.warp(X), self.warp(T)
        nt = T.shape[0]
        covs = np.empty(X.shape[0])
        step = max(1, 2048 - nt)
        for lo in range(0, X.shape[0], step):
            Q = np.vstack([T, X[lo : lo + step]])
            kss = k.diag(Q)
            _mean, _var, C = self._ctx.predict_gram(k(Q, Xt)[None], kss[None], k(Q)[None])
            C = C[0]
            tt = kss[:nt] - np.diag(C)[:nt]
            cross = C[:nt, nt:]
            cii = np.diag(C)[nt:]
            covs[lo : lo + step] = tt.sum() + np.sum(cross * cross / cii[None, :], axis=0)
        return covs

    def _mvn_mode(self, m, mvn=None):
        """'reference' (numpy's SVD draw on the host from the device-built mean / covariance) or 'cholesky' (device)."""
        mode = self.__dict__.get("mvn", "auto") if mvn is None else mvn
        if mode not in ("auto", "reference", "cholesky"):
            raise ValueError("mvn must be 'auto', 'reference' or 'cholesky', got %r" % (mode,))
        if mode == "auto":
            mode = "reference" if m <= self.MVN_REFERENCE_MAX_POINTS else "cholesky"
        if self._generic:
            # the device draw builds the predictive covariance from the canonical hyper-parameters; a generic tree has none:
            # mean / covariance through the host-evaluated kernel, the reference's own SVD draw on the host
            mode = "reference"
        return mode

    def sample_y(self, X, sample_mean=False, noise=False, n_samples=1, random_state=0, mvn=None):
        """Function realisations of the GP(s) (``bask/bayesgpr.py:637-718``).  Predictive means and
        covariances are built on the device; the multivariate normal draw is numpy's legacy SVD draw on the
        host (``mvn="reference"``: the reference's own variates, ``sklearn/_gpr.py:522-526``) or
        ``mean + chol(cov) z`` on the device (``mvn="cholesky"``: same distribution, other variates).
        ``mvn=None`` takes the estimator's setting (default ``"auto"``: reference variates up to
        ``MVN_REFERENCE_MAX_POINTS`` query points)."""
        rng = check_random_state(random_state)
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        mode = self._mvn_mode(X.shape[0], mvn)
        if sample_mean:
            cm = nullcontext(self) if noise else self.noise_set_to_zero()
            with cm:
                if mode == "reference":
                    # sklearn's sample_y: predict(return_cov=True), then rng.multivariate_normal(mean, cov, n).T
                    y_mean, y_cov = self.predict(X, return_cov=True)
                    return _legacy_mvn(rng, y_mean, y_cov, n_samples).T
                return self._draw(X, n_samples, rng)
        # hyper-posterior draws: one chain row (with replacement) and one function per sample.  The generator is
        # consumed in the reference's order -- all row indices first, then one normal vector per sample -- and the
        # device builds all posteriors and predictive covariances in batched calls.
        ind = rng.choice(len(self.chain_), size=n_samples, replace=True)
        if mode == "reference":
            return self._draw_rows_reference(self.chain_[ind], X, rng, noise).T
        Z = np.vstack([rng.standard_normal((1, X.shape[0])) for _ in range(n_samples)])
        return self._draw_rows(self.chain_[ind], X, Z, noise).T

    def _sample_hyper_rows(self, n_draws, X, rng):
        """The sample acquisitions of ``evaluate_acquisitions``: per draw the reference calls
        ``gpr.sample_y(X, random_state=rng)`` (``bask/acquisition.py:132-136``), i.e. ONE function of ONE freshly
        chosen chain row with the noise off -- same generator consumption here (row index, then the normal vector,
        per draw), one batched device call for all draws.  Returns (n_draws, m)."""
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        if self._mvn_mode(X.shape[0]) == "reference":
            # row index and MVN draw alternate on the generator, as in the reference's loop: the rows cannot be chosen
            # ahead of the draws, so every draw is one device predict (mean, covariance) + one host SVD
            out = np.empty((n_draws, X.shape[0]))
            for i in range(n_draws):
                row = self.chain_[rng.choice(len(self.chain_), size=1, replace=True)[0]]
                out[i] = self._draw_rows_reference(row[None, :], X, rng, noise=False)[0]
            return out
        rows, Z = [], []
        for _ in range(n_draws):
            rows.append(self.chain_[rng.choice(len(self.chain_), size=1, replace=True)[0]])
            Z.append(rng.standard_normal((1, X.shape[0])))
        return self._draw_rows(np.array(rows), X, np.vstack(Z), noise=False)

    def _draw_rows_reference(self, rows, X, rng, noise):
        """The reference's per-sample loop (``bask/bayesgpr.py:679-706``) with the device doing what the theta setter
        and ``predict(return_cov=True)`` do there -- ONE batched posterior build over the distinct chain rows, the
        predictive means / covariances in chunks -- and the host doing what numpy does there: one legacy
        ``multivariate_normal`` (SVD) per sample, in sample order, on the caller's generator.  (len(rows), m)."""
        rows = np.atleast_2d(rows)
        n_theta = len(self.kernel_.theta)
        m = X.shape[0]
        out = np.empty((len(rows), m))
        if self.warp_inputs:
            validate_zeroone(X)
            d = self._X_train_.shape[1]
            backup = (np.copy(self.warp_alphas_), np.copy(self.warp_betas_))
            try:
                for i, row in enumerate(rows):  # every draw has its own warped training inputs
                    self.create_warpers(row[n_theta : n_theta + d], row[n_theta + d :])
                    self.rewarp()
                    mean, cov = self._mean_cov_rows(row[None, :n_theta], X, noise)
                    out[i] = _legacy_mvn(rng, mean[0], cov[0], 1)[0]
            finally:
                self.create_warpers(*backup)
                self.rewarp()
            return out
        # the covariances of a chunk of samples at a time (m^2 doubles each)
        chunk = max(1, int((256 << 20) // (8 * m * m)))
        for lo in range(0, len(rows), chunk):
            mean, cov = self._mean_cov_rows(rows[lo : lo + chunk, :n_theta], X, noise)
            for i in range(mean.shape[0]):
                out[lo + i] = _legacy_mvn(rng, mean[i], cov[i], 1)[0]
        return out

    def _mean_cov_rows(self, thetas, X, noise):
        """Predictive mean and covariance (y units) of the GP of every chain row: batched device posterior build over the
        distinct rows + batched device predict with the full covariance; ``theta`` / ``alpha_`` / ``L_`` stay untouched."""
        mean, _var, cov = self._rows_predict(np.atleast_2d(thetas), X, noise_zero=not noise, return_cov=True)
        return self.y_train_std_ * mean + self.y_train_mean_, cov * self.y_train_std_**2

    def _draw_rows(self, rows, X, Z, noise):
        """f_i = mean_i + chol(cov_i) Z[i] for the GP of chain row i (kernel parameters and, with input warping, its
        own warp); (len(rows), m) in the units of y.  The resident posterior of ``theta`` is rebuilt lazily by the
        next call that needs it (``_make_resident``), so ``theta`` / ``alpha_`` / ``L_`` / ``K_inv_`` are untouched."""
        rows = np.atleast_2d(rows)
        n_theta = len(self.kernel_.theta)
        if self.warp_inputs:
            validate_zeroone(X)
            d = self._X_train_.shape[1]
            backup = (np.copy(self.warp_alphas_), np.copy(self.warp_betas_))
            out = np.empty((len(rows), X.shape[0]))
            try:
                for i, row in enumerate(rows):  # every draw has its own warped training inputs
                    self.create_warpers(row[n_theta : n_theta + d], row[n_theta + d :])
                    self.rewarp()
                    out[i] = self._draw_rows_device(row[None, :n_theta], X, Z[i : i + 1], noise)[0]
            finally:
                self.create_warpers(*backup)
                self.rewarp()
            return out
        return self._draw_rows_device(rows[:, :n_theta], X, Z, noise)

    def _draw_rows_device(self, thetas, X, Z, noise):
        H = self._canonical(thetas)
        uniq, inverse = np.unique(H, axis=0, return_inverse=True)  # repeated chain rows share one posterior build
        res = self._ctx.posterior(uniq, want_alpha=False)
        if np.any(res["status"] != 0):
            bad = int(np.flatnonzero(res["status"])[0])
            raise np.linalg.LinAlgError(
                _PD_MESSAGE % self.kernel_,
                "%d-th leading minor of the array is not positive definite" % res["status"][bad],
            )
        Hk = H.copy()
        if not noise:
            Hk[:, -1] = -np.inf  # noise_set_to_zero(): the factors keep the noise, the predictive kernel drops it
        pidx = np.asarray(inverse, dtype=np.int32).ravel()
        out = np.empty_like(Z)
        todo = np.arange(len(H))
        jitter = 1e-10
        while len(todo):
            o, st = self._ctx.sample_y_batch(pidx[todo], Hk[todo], X, Z[todo], jitter=jitter)
            ok = st == 0
            out[todo[ok]] = o[ok]
            todo = todo[~ok]
            # numpy's SVD-based draw tolerates a numerically semi-definite covariance; the Cholesky-based draw
            # needs a growing diagonal jitter instead
            jitter *= 100.0
            if len(todo) and jitter > 1e-2:
                raise _lib.NotPositiveDefinite("predictive covariance not positive definite (jitter %.3g)" % jitter)
        return self.y_train_std_ * out + self.y_train_mean_

    def _draw(self, X, n_samples, rng):
        self._make_resident()
        Hk = self._canonical(self._kernel_theta_for_predict())
        z = rng.standard_normal((n_samples, X.shape[0]))
        jitter = 1e-10
        while True:
            try:
                out = self._ctx.sample_y(0, Hk, X, z, jitter=jitter)
                break
            except _lib.NotPositiveDefinite:
                # numpy's SVD-based draw tolerates a numerically semi-definite covariance; the
                # Cholesky-based draw needs a growing diagonal jitter instead
                jitter *= 100.0
                if jitter > 1e-2:
                    raise
        return (self.y_train_std_ * out + self.y_train_mean_).T

    def __del__(self):
        ctx = self.__dict__.get("_ctx_obj")
        if ctx is not None:
            try:
                ctx.close()
            except Exception:
                pass


def _with_white_zeroed(kernel):
    """A copy of ``kernel`` with its WhiteKernel (inside nested sums) at level 0: what ``noise_set_to_zero`` does to
    ``kernel_`` (``bask/bayesgpr.py:327-333``)."""
    k = clone(kernel)
    if isinstance(k, WhiteKernel):
        k.set_params(noise_level=0.0)
        return k
    present, white_param = param_for_white_kernel_in_sum(k)
    if present:
        k.set_params(**{white_param: WhiteKernel(noise_level=0.0)})
    return k


def _legacy_mvn(rng, mean, cov, n_samples):
    """``rng.multivariate_normal(mean, cov, n_samples)`` -- numpy's legacy SVD-based draw, the call scikit-learn's
    ``sample_y`` makes (``sklearn/_gpr.py:522-526``); (n_samples, m).  Its "covariance is not symmetric positive-
    semidefinite" warning is the reference's too (noise-free predictive covariances are numerically singular)."""
    return np.atleast_2d(rng.multivariate_normal(np.ravel(mean), cov, n_samples))


def _eval_warp_priors(warp_priors, W, d):
    """Log-prior of the warp parameters (``bask/bayesgpr.py:360-365``): a pair of callables applied to
    every (alpha_k, beta_k), or one callable of both."""
    lp = np.zeros(W.shape[0])
    A, Bm = W[:, :d], W[:, d:]
    if isinstance(warp_priors, (list, tuple)):
        for k in range(d):
            lp += _vec_call(warp_priors[0], A[:, k]) + _vec_call(warp_priors[1], Bm[:, k])
    else:
        for k in range(d):
            lp += np.array([float(warp_priors(a, b)) for a, b in zip(A[:, k], Bm[:, k])])
    return lp


class _ShardedLogProb:
    """``log_prob_fn`` of the exact single-ensemble sharding (SURVEY.md 8e option 1; the reference runs ONE ensemble on
    one RNG, ``bask/bayesgpr.py:490-530``): every rank is handed the same (B, p) block, evaluates the LML of its own
    rows [r B/G, (r+1) B/G) on its device and the log-priors of ALL rows on the host meanwhile; ``finish`` gathers the B
    log-likelihoods device to device (``distributed.allgather_lml``: one RCCL all-gather of B doubles per half-step).
    Same arithmetic per row as ``_AsyncLogProb``: the chain equals the single-GPU chain bit for bit."""

    def __init__(self, gp):
        self._gp = gp

    def __call__(self, Theta, priors, warp_priors=None):
        return self.finish(self.begin(Theta, priors, warp_priors))

    def begin(self, Theta, priors, warp_priors=None):
        gp = self._gp
        Theta = np.atleast_2d(np.asarray(Theta, dtype=np.float64))
        if distributed.backend() is None:
            return ("single", gp._log_prob_begin(Theta, priors, warp_priors))
        rank, _lr, ws = distributed.world()
        B = Theta.shape[0]
        lo, hi = distributed.shard_rows(B, rank, ws)
        # From here on every rank is committed to ONE collective in ``finish``.  Whatever fails locally in between -- the
        # priors raising, a failed submit, a device error -- is carried in the token and reported THROUGH that collective
        # (status word next to the values): a rank that raised here would leave its peers blocked in the all-gather.
        if gp.warp_inputs or gp._generic:  # per-walker warps / host-evaluated kernels: finished values gathered from the host
            try:
                return ("host", B, gp._log_prob_begin(Theta[lo:hi], priors, warp_priors) if hi > lo else None, None)
            except Exception as exc:
                return ("host", B, None, exc)
        # decided from what every rank sees alike (same block, same context settings): all ranks gather the same way
        can_async = not gp._ctx._timing and -(-B // ws) <= gp._ctx.max_batch
        native = distributed.backend() == "rccl" and can_async
        submitted, lp, H, failure = False, None, None, None
        try:
            H = gp._canonical(Theta[lo:hi])
            submitted = gp._ctx.lml_submit(H) if (can_async and hi > lo) else False
            lp = _eval_priors(priors, Theta)  # all rows on every rank, while the device factorises this rank's
        except Exception as exc:
            failure = exc
        except BaseException:
            # KeyboardInterrupt / SystemExit: this rank is going down and will never enter the collective.  Collect the pending
            # batch (otherwise every later call on the context answers BGP_ERR_STATE) and take the group down with it, so that the
            # peers fail at once instead of sitting in the all-gather until BGP_COMM_TIMEOUT_S.
            try:
                if submitted or gp._ctx.has_pending():
                    gp._ctx.lml_wait()
            except Exception:
                pass
            distributed.abort_process_group()
            raise
        return ("dev", B, lp, H, submitted, native, hi > lo, failure)

    def finish(self, token):
        gp = self._gp
        if token[0] == "single":
            return gp._log_prob_finish(token[1])
        if token[0] == "host":
            _k, B, tok, failure = token
            local = np.zeros(0)
            if failure is None and tok is not None:
                try:
                    local = gp._log_prob_finish(tok)
                except Exception as exc:
                    failure = exc
            return self._gather(None, B, local, failure)
        _k, B, lp, H, submitted, native, has_rows, failure = token
        if native:
            lml = self._gather(gp._ctx, B, None, failure)
        else:  # gloo group (CPU tests, ranks sharing one GPU) or a batch that could not go asynchronously
            local = np.zeros(0)
            try:
                if submitted:
                    local = gp._ctx.lml_wait()  # (also when the priors failed: the pending batch must be collected)
                elif has_rows and failure is None:
                    local = gp._ctx.lml(H)
            except Exception as exc:
                failure = failure or exc
            lml = self._gather(None, B, local, failure)
        with np.errstate(invalid="ignore"):
            lp = lp + lml
        lp[~np.isfinite(lp)] = -np.inf
        return lp

    def resident(self, n_walkers, n_dim, priors=None, warp_priors=None):
        """The sharded ensemble's run resident on every rank's device (``bgp_mcmc_begin_ex`` with the group's communicator):
        the step kernel replicated, each rank's rows of a half-step factorised on its GPU, the all-gather of the
        log-likelihoods ON THE STREAM between the batch and the next step kernel -- no host synchronisation per half-step
        (``bask/bayesgpr.py:510-530`` on G GPUs).  Over gloo (CPU tests, ranks sharing a GPU) the host-driven exchange stays."""
        if distributed.backend() is None:
            run, self.resident_reason = _resident_run(self._gp, n_walkers, n_dim, priors, warp_priors, None)
            return run
        if distributed.backend() != "rccl":
            run, self.resident_reason = None, "the process group is a gloo group: the exchange goes through the host"
            return run
        run, self.resident_reason = _resident_run(self._gp, n_walkers, n_dim, priors, warp_priors, distributed.communicator())
        return run

    @staticmethod
    def _gather(ctx, B, local, failure):
        try:
            return distributed.allgather_lml(ctx, B, local=local, error=0 if failure is None else 1)
        except distributed.ShardedEvaluationError as err:
            if failure is not None:
                raise err from failure
            raise


class _AsyncLogProb:
    """``log_prob_fn`` of the ensemble sampler: callable like ``_log_prob_batch`` and, for the sampler's overlap of
    its own bookkeeping with the device, split into ``begin`` (enqueue) / ``finish`` (collect)."""

    def __init__(self, gp):
        self._gp = gp

    def __call__(self, Theta, priors, warp_priors=None):
        return self._gp._log_prob_batch(Theta, priors, warp_priors)

    def begin(self, Theta, priors, warp_priors=None):
        return self._gp._log_prob_begin(Theta, priors, warp_priors)

    def finish(self, token):
        return self._gp._log_prob_finish(token)

    def resident(self, n_walkers, n_dim, priors=None, warp_priors=None):
        """The sampler asks: can the whole run stay on the device (``bgp_mcmc_begin`` / ``_steps`` / ``_end``)?  An object
        with begin(coords, log_prob, nsteps) / steps(plan segment) / progress() / end() / abandon() when it can, else None with
        the reason in ``resident_reason`` (the host-driven loop runs and says so once)."""
        run, self.resident_reason = _resident_run(self._gp, n_walkers, n_dim, priors, warp_priors, None)
        return run


def _device_prior(fn):
    """(kind, five parameters) of a log-prior the step kernel of the resident sampler can evaluate (``include/bgp.h``,
    ``bgp_mcmc_begin_ex``), or None: the two families ``guess_priors`` hands out carry theirs (``_bgp_device``); a frozen
    ``scipy.stats.norm(loc, scale).logpdf`` -- the reference's default warp priors, ``bask/bayesgpr.py:463-466`` -- is kind 3."""
    dev = getattr(fn, "_bgp_device", None)
    if dev is not None:
        return dev
    frozen = getattr(fn, "__self__", None)
    dist = getattr(frozen, "dist", None)
    if getattr(fn, "__name__", "") != "logpdf" or getattr(dist, "name", None) != "norm":
        return None
    try:
        args, loc, scale = dist._parse_args(*frozen.args, **frozen.kwds)
        loc, scale = float(loc), float(scale)
    except Exception:
        return None
    if args or not (np.isfinite(loc) and np.isfinite(scale) and scale > 0.0):
        return None
    # scipy: x = (t - loc) / scale;  -x**2 / 2.0 - log(sqrt(2 pi)) - log(scale), with numpy's constants
    return 3, (loc, scale, float(np.log(np.sqrt(2 * np.pi))), float(np.log(np.asarray(scale))), 0.0)


def _resident_run(gp, n_walkers, n_dim, priors, warp_priors, comm):
    """(run object, None) when the ensemble sampler's whole run can stay on the device, (None, reason) otherwise.
    ``comm``: the communicator of a sharded ensemble (every rank decides from the same facts: the same answer everywhere)."""
    if not getattr(gp, "resident_sampler", True):
        return None, None  # (asked for: not a fallback)
    if gp._generic:
        return None, "the kernel tree has no canonical device form: its matrices are evaluated on the host"
    if n_walkers % 2:
        return None, "an odd number of walkers: the two halves of a step differ in size"
    ctx = gp._ctx
    if ctx._timing:
        return None, "per-launch timing is on"
    Ns = n_walkers // 2
    world = comm.world if comm is not None else 1
    if -(-Ns // world) > ctx.max_batch:
        return None, "a half-step's proposals exceed max_batch"
    if callable(priors) or priors is None:
        return None, "the prior is one callable of the whole parameter vector"
    priors = list(priors)
    d = gp._X_train_.shape[1]
    nwarp = 2 * d if gp.warp_inputs else 0
    n_theta = n_dim - nwarp
    if len(priors) != n_theta:
        return None, "the number of priors differs from the number of kernel hyper-parameters"
    table = [_device_prior(f) for f in priors]
    if nwarp:
        if not isinstance(warp_priors, (list, tuple)) or len(warp_priors) != 2:
            return None, "the warp prior is one callable of (alpha, beta)"
        wa, wb = _device_prior(warp_priors[0]), _device_prior(warp_priors[1])
        table += [wa] * d + [wb] * d
    if any(t is None for t in table):
        return None, "a prior is not one of the families the device evaluates (guess_priors' two, scipy.stats.norm)"
    try:  # the canonical map as an index table: probe it with the positions themselves
        probe = gp._canonical(np.arange(n_theta, dtype=np.float64)[None, :] + 0.25)[0]
        fixed = gp._canonical(np.arange(n_theta, dtype=np.float64)[None, :] + 0.75)[0]
    except Exception:
        return None, "the kernel's hyper-parameters do not map onto the canonical vector entry by entry"
    src = np.where(probe != fixed, np.floor(probe).astype(np.int64), -1)
    if probe.shape != (d + 2,) or np.any(src >= n_theta):
        return None, "the kernel's hyper-parameters do not map onto the canonical vector entry by entry"
    kind = np.array([t[0] for t in table], dtype=np.int32)
    par = np.array([t[1] for t in table], dtype=np.float64)
    h_fixed = np.where(src < 0, probe, 0.0)

    class _Run:  # (the sampler hands the plan over in segments and draws the next one while the device works)
        @staticmethod
        def begin(coords, log_prob, nsteps):
            ctx.mcmc_begin(coords, log_prob, nsteps, src, h_fixed, kind, par, comm=comm, nwarp=nwarp)

        steps = staticmethod(ctx.mcmc_steps)
        progress = staticmethod(ctx.mcmc_progress)
        end = staticmethod(ctx.mcmc_end)
        abandon = staticmethod(ctx.mcmc_abandon)

    return _Run, None


def _vec_call(fn, col):
    try:
        with np.errstate(all="ignore"):
            v = np.asarray(fn(col), dtype=np.float64)
        if v.shape == col.shape:
            return v
    except Exception:
        pass
    with np.errstate(all="ignore"):
        return np.array([float(fn(t)) for t in col])


def _eval_priors(priors, Theta):
    """Sum of log-priors for a (Ns, p) block.  A list holds one callable per hyper-parameter
    (``bask/bayesgpr.py:368-370``), evaluated column-wise when the callable broadcasts and
    element-wise otherwise; a single callable receives each full parameter vector (:371-372)."""
    Ns, p = Theta.shape
    if callable(priors):
        return np.array([float(priors(row)) for row in Theta])
    priors = list(priors)
    if len(priors) != p:
        raise ValueError(f"zip() argument 2 is {'shorter' if p < len(priors) else 'longer'} than argument 1: "
                         f"{len(priors)} priors for {p} hyper-parameters")
    lp = np.zeros(Ns)
    k = 0
    while k < p:
        prior = priors[k]
        k1 = k + 1
        while k1 < p and priors[k1] is prior:  # guess_priors puts the SAME callable on every length scale
            k1 += 1
        block = Theta[:, k:k1]
        vals = None
        try:  # one elementwise call for the whole run of columns ...
            with np.errstate(all="ignore"):
                v = np.asarray(prior(block if k1 - k > 1 else block[:, 0]), dtype=np.float64)
            if v.shape == block.shape or (k1 - k == 1 and v.shape == (Ns,)):
                vals = v.reshape(Ns, k1 - k)
        except Exception:
            vals = None
        if vals is None:
            with np.errstate(all="ignore"):
                vals = np.array([[float(prior(t)) for t in row] for row in block])
        for j in range(k1 - k):  # ... added column by column, i.e. in the reference's summation order
            lp += vals[:, j]
        k = k1
    return lp
