"""Stepwise Bayesian optimisation: ``Optimizer.ask / tell / run`` (host mirror of
``bask/optimizer.py:35-445``).  ``tell`` is the per-iteration entry of the hot path: it
(re)fits / resumes the BayesGPR hyper-posterior MCMC on the device and evaluates the acquisition
function over ``n_points`` random candidates with the device predict / PVRS kernels.

The post-hoc diagnostics ``probability_of_optimality`` / ``expected_optimality_gap`` /
``optimum_intervals`` (``bask/optimizer.py:447-689``) run on the device ``sample_y``.
"""
import warnings

import numpy as np
from sklearn.utils import check_random_state

from . import acquisition
from .acquisition import evaluate_acquisitions
from .bayesgpr import BayesGPR
from .init import r2_sequence, sb_sequence
from .space import create_result, is_2Dlistlike, is_listlike, normalize_dimensions
from .utils import construct_default_kernel, expected_minimum, hdi

__all__ = ["Optimizer"]

ACQUISITION_FUNC = {
    "ei": acquisition.ExpectedImprovement(),
    "lcb": acquisition.LCB(),
    "mean": acquisition.Expectation(),
    "mes": acquisition.MaxValueSearch(),
    "pvrs": acquisition.PVRS(),
    "ts": acquisition.ThompsonSampling(),
    "ttei": acquisition.TopTwoEI(),
    "vr": acquisition.VarianceReduction(),
}

_INT32_MAX = np.iinfo(np.int32).max


class Optimizer:
    """Ask/tell optimiser with the constructor arguments, public attributes (``rng, space, gp, gp_priors, Xi, yi,
    noisei, n_points, n_initial_points_, init_strategy, acq_func, acq_func_kwargs``) and defaults of
    ``bask/optimizer.py:120-175``.  The generator ``rng`` is consumed in the reference's order (initial-design seed,
    surrogate seed; per tell: candidate draw, acquisition seed), so a seeded run visits the same points."""

    def __init__(self, dimensions, n_points=500, n_initial_points=10, init_strategy="sb", gp_kernel=None,
                 gp_kwargs=None, gp_priors=None, acq_func="pvrs", acq_func_kwargs=None, random_state=None, **kwargs):
        self.rng = check_random_state(random_state)
        self.space = normalize_dimensions(dimensions)
        self.n_points = n_points
        self.acq_func = acq_func if callable(acq_func) else ACQUISITION_FUNC[acq_func]
        self.acq_func_kwargs = acq_func_kwargs if acq_func_kwargs is not None else {}
        self._plan_initial_design(n_initial_points, init_strategy)
        self.gp = self._new_surrogate(gp_kernel, gp_kwargs)
        self.gp_priors = gp_priors
        self._forget_observations()
        self._next_x = None
        self._last_candidates = None  # transformed candidates / acquisition values of the latest proposal (tests, plots)
        self._last_acq_values = None

    # ---- construction helpers ------------------------------------------------------------------------------
    def _plan_initial_design(self, n_initial_points, init_strategy):
        """"r2": the whole quasi-random design up front; "sb": a private generator for the sequential Steinerberger
        points (seeded from ``rng``); anything else: uniform random points drawn on demand."""
        self.n_initial_points_ = self._n_initial_points = n_initial_points
        self.init_strategy = init_strategy
        if init_strategy == "r2":
            self._initial_points = self.space.inverse_transform(r2_sequence(n=n_initial_points, d=self.space.n_dims))
        elif init_strategy == "sb":
            self._init_rng = np.random.RandomState(self.rng.randint(2**31))

    def _new_surrogate(self, kernel, gp_kwargs):
        if kernel is None:  # the default kernel only needs the number of (transformed) dimensions
            kernel = construct_default_kernel(list(range(self.space.transformed_n_dims)))
        return BayesGPR(kernel=kernel, random_state=self.rng.randint(0, _INT32_MAX), **(gp_kwargs or {}))

    def _forget_observations(self):
        self.Xi, self.yi, self.noisei = [], [], []

    def _result(self):
        return create_result(self.Xi, self.yi, self.space, self.rng, models=[self.gp])

    # ---- ask ---------------------------------------------------------------------------------------------------
    def ask(self, n_points=1):
        """Next point to evaluate (``bask/optimizer.py:177-226``): a point of the initial design while it lasts, then
        the maximiser of the acquisition function found by the last ``tell``."""
        if n_points > 1:
            raise NotImplementedError("Returning multiple points is not implemented yet.")
        if self._n_initial_points <= 0:
            if not self.gp.kernel_:
                raise RuntimeError("Initialization is finished, but no model has been fit.")
            return self._next_x
        if self.init_strategy == "r2":
            return self._initial_points[self._n_initial_points - 1]
        if self.init_strategy != "sb":
            return self.space.rvs()[0]
        seen = len(self.Xi)
        design = sb_sequence(n=seen + 1, d=self.space.transformed_n_dims,
                             existing_points=self.space.transform(self.Xi) if seen else None,
                             random_state=self._init_rng.randint(2**31))
        return self.space.inverse_transform(np.atleast_2d(design[seen]))[0]

    # ---- tell --------------------------------------------------------------------------------------------------
    def _record(self, x, y, noise_vector):
        """Append one observation (x a point, y a number) or a batch (x a list of points, y a list); returns how many
        were added.  Noise variances default to 0 and must match the shape of y (``bask/optimizer.py:298-329``)."""
        batch = is_listlike(y) and is_2Dlistlike(x)
        if not batch and not is_listlike(x):
            raise ValueError(f"Type of arguments `x` ({type(x)}) and `y` ({type(y)}) not compatible.")
        if batch:
            if noise_vector is None:
                noise_vector = [0.0] * len(y)
            elif not is_listlike(noise_vector) or len(noise_vector) != len(y):
                raise ValueError("Vector of noise variances needs to be of equal length as `y`.")
            xs, ys, noises = list(x), list(y), list(noise_vector)
        else:
            if is_listlike(noise_vector):
                raise ValueError("Vector of noise variances is a list, while tell only received one datapoint.")
            xs, ys, noises = [x], [y], [0.0 if noise_vector is None else noise_vector]
        self.Xi += xs
        self.yi += ys
        self.noisei += noises
        return len(ys)

    def _update_surrogate(self, from_scratch, gp_samples, gp_burnin, progress):
        """First model (or after ``replace``): MAP fit + MCMC (``BayesGPR.fit``); afterwards the walkers resume from
        their last positions on the grown data set (``BayesGPR.sample``) -- ``bask/optimizer.py:330-351``."""
        if self.gp_priors is not None and len(self.gp_priors) != self.space.transformed_n_dims + 2:
            raise ValueError("The number of priors does not match the number of dimensions + 2.")
        infer = self.gp.fit if (from_scratch or self.gp.pos_ is None) else self.gp.sample
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            infer(self.space.transform(self.Xi), self.yi, noise_vector=np.array(self.noisei), priors=self.gp_priors,
                  n_desired_samples=gp_samples, n_burnin=gp_burnin, progress=progress)

    def _propose(self, n_samples):
        """Maximise the acquisition function over ``n_points`` random candidates (uniform in the warped space when
        the surrogate warps its inputs, ``bask/optimizer.py:353-357``) on the device."""
        if self.gp.warp_inputs:
            cand = self.gp.unwarp(self.rng.uniform(size=(self.n_points, self.space.transformed_n_dims)))
        else:
            cand = self.space.rvs_transformed(n_samples=self.n_points, random_state=self.rng)
        values = evaluate_acquisitions(X=cand, gpr=self.gp, acquisition_functions=(self.acq_func,), n_samples=n_samples,
                                       progress=False, random_state=self.rng.randint(0, _INT32_MAX),
                                       **self.acq_func_kwargs).ravel()
        self._last_candidates, self._last_acq_values = cand, values
        return self.space.inverse_transform(cand[np.argmax(values)].reshape((1, -1)))[0]

    def tell(self, x, y, noise_vector=None, fit=True, replace=False, n_samples=0, gp_samples=100, gp_burnin=10,
             progress=False):
        """Record observation(s); once the initial design is used up (and ``fit``) update the hyper-posterior MCMC of
        the surrogate and choose the next point.  Arguments, return value (an ``OptimizeResult`` with ``x, fun,
        x_iters, func_vals, space, models``) and raised errors as ``bask/optimizer.py:228-380``."""
        if replace:
            self._forget_observations()
            self._n_initial_points = self.n_initial_points_
        self._n_initial_points -= self._record(x, y, noise_vector)
        if fit and self._n_initial_points <= 0:
            self._update_surrogate(replace, gp_samples, gp_burnin, progress)
            self._next_x = self._propose(n_samples)
        return self._result()

    def run(self, func, n_iter=1, replace=False, n_samples=5, gp_samples=100, gp_burnin=10):
        """``n_iter`` rounds of ask -> ``func`` -> tell; ``func`` returns the objective value or a (value, noise
        variance) pair; ``replace`` applies to the first round only (``bask/optimizer.py:382-445``)."""
        for it in range(n_iter):
            x = self.ask()
            outcome = func(x)
            value, noise = outcome if hasattr(outcome, "__len__") else (outcome, 0.0)
            self.tell(x, value, noise_vector=noise, replace=replace and it == 0, n_samples=n_samples,
                      gp_samples=gp_samples, gp_burnin=gp_burnin)
        return self._result()

    def _optimum_vs_space_draws(self, n_space_samples, n_gp_samples, n_random_starts, use_mean_gp, seed):
        """Function draws (device ``sample_y``) at [expected optimum, n_space_samples random points]:
        (1 + n_space_samples, n_gp_samples); row 0 belongs to the minimiser of the surrogate mean found by
        ``utils.expected_minimum`` (``bask/optimizer.py:493-512``).  ``seed`` is handed unchanged to each of the
        three consumers, as the reference does."""
        res = self._result()
        x_opt, _ = expected_minimum(res, n_random_starts=n_random_starts, random_state=seed)
        points = [x_opt] + self.space.rvs(n_samples=n_space_samples, random_state=seed)
        return self.gp.sample_y(self.space.transform(points), sample_mean=use_mean_gp, n_samples=n_gp_samples,
                                random_state=seed)

    def probability_of_optimality(self, threshold, n_space_samples=500, n_gp_samples=200, n_random_starts=100,
                                  use_mean_gp=True, normalized_scores=True, random_state=None):
        """Probability that no point of the space beats the current expected optimum by more than
        ``threshold`` (a float, or a list giving a list of probabilities) -- ``bask/optimizer.py:447-525``,
        same arguments.  With ``normalized_scores`` the gaps are measured in units of each draw's standard
        deviation over the points."""
        draws = self._optimum_vs_space_draws(n_space_samples, n_gp_samples, n_random_starts, use_mean_gp, random_state)
        gap = draws[0][None, :] - draws          # how much better every point is than the optimum, per draw
        if normalized_scores:
            gap = gap / np.std(draws, axis=0)
        worst = gap.max(axis=0)                  # the best competitor in each draw
        probs = [float(np.mean(worst - eps < 0.0)) for eps in (threshold if is_listlike(threshold) else [threshold])]
        return probs if is_listlike(threshold) and len(probs) > 1 else probs[0]

    def expected_optimality_gap(self, max_tries=3, n_probabilities=50, n_space_samples=500, n_gp_samples=200,
                                n_random_starts=100, tol=0.01, use_mean_gp=True, normalized_scores=True,
                                random_state=None):
        """Expected optimality gap of the current global optimum (``bask/optimizer.py:527-620``): the
        distribution function of the gap is sampled with ``probability_of_optimality`` on ``n_probabilities``
        thresholds between 0 and the smallest threshold at which the probability reaches 1 (found by a
        bounded scalar minimisation of ``(p - 1)^2 + 1e-3 t^2``, at most ``max_tries`` attempts)."""
        from scipy.optimize import minimize_scalar

        seed = check_random_state(random_state).randint(0, 2**32 - 1, dtype=np.int64)
        kw = dict(n_space_samples=n_space_samples, n_gp_samples=n_gp_samples, n_random_starts=n_random_starts,
                  use_mean_gp=use_mean_gp, normalized_scores=normalized_scores, random_state=seed)
        span = float(np.max(self.yi) - np.min(self.yi))
        upper = None
        for _ in range(max_tries):
            try:
                upper = minimize_scalar(
                    lambda t: (self.probability_of_optimality(threshold=t, **kw) - 1.0) ** 2 + 1e-3 * t * t,
                    bounds=(0.0, span), tol=tol).x
            except ValueError:
                continue
            break
        if upper is None:
            raise ValueError("Determining the upper threshold was not possible.")
        grid = np.linspace(0.0, upper, num=n_probabilities)
        cdf = np.asarray(self.probability_of_optimality(list(grid), **kw), dtype=np.float64)
        return float(np.sum(np.diff(cdf) * grid[1:]))

    def optimum_intervals(self, hdi_prob=0.95, multimodal=True, opt_samples=200, space_samples=500, only_mean=True,
                          random_state=None):
        """Highest density intervals of the optimum's location per dimension by Thompson sampling
        (``bask/optimizer.py:622-689``); ``utils.hdi`` restates the two arviz estimators."""
        if self.space.is_partly_categorical:
            raise NotImplementedError("Highest density interval not implemented for categorical parameters.")
        X = self.space.transform(self.space.rvs(n_samples=space_samples, random_state=random_state))
        optimum_samples = self.gp.sample_y(X, sample_mean=only_mean, n_samples=opt_samples, random_state=random_state)
        X_opt = X[np.argmin(optimum_samples, axis=0)]
        intervals = []
        for i, col in enumerate(X_opt.T):
            raw_interval = hdi(col, hdi_prob=hdi_prob, multimodal=multimodal)
            intervals.append(self.space.dimensions[i].inverse_transform(raw_interval))
        return intervals
