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


class Optimizer:
    """Constructor arguments, attributes and defaults as ``bask/optimizer.py:120-133``."""

    def __init__(
        self,
        dimensions,
        n_points=500,
        n_initial_points=10,
        init_strategy="sb",
        gp_kernel=None,
        gp_kwargs=None,
        gp_priors=None,
        acq_func="pvrs",
        acq_func_kwargs=None,
        random_state=None,
        **kwargs,
    ):
        self.rng = check_random_state(random_state)
        self.acq_func = acq_func if callable(acq_func) else ACQUISITION_FUNC[acq_func]
        self.acq_func_kwargs = {} if acq_func_kwargs is None else acq_func_kwargs

        self.space = normalize_dimensions(dimensions)
        self._n_initial_points = n_initial_points
        self.n_initial_points_ = n_initial_points
        self.init_strategy = init_strategy
        if self.init_strategy == "r2":
            self._initial_points = self.space.inverse_transform(r2_sequence(n=n_initial_points, d=self.space.n_dims))
        elif self.init_strategy == "sb":
            self._init_rng = np.random.RandomState(self.rng.randint(2**31))
        self.n_points = n_points

        if gp_kwargs is None:
            gp_kwargs = {}
        if gp_kernel is None:
            gp_kernel = construct_default_kernel(list(range(self.space.transformed_n_dims)))
        self.gp = BayesGPR(kernel=gp_kernel, random_state=self.rng.randint(0, np.iinfo(np.int32).max), **gp_kwargs)
        self.gp_priors = gp_priors

        self.Xi = []
        self.yi = []
        self.noisei = []
        self._next_x = None

    def ask(self, n_points=1):
        """Next point to evaluate (``bask/optimizer.py:177-226``)."""
        if n_points > 1:
            raise NotImplementedError("Returning multiple points is not implemented yet.")
        if self._n_initial_points > 0:
            if self.init_strategy == "r2":
                return self._initial_points[self._n_initial_points - 1]
            if self.init_strategy == "sb":
                existing = self.space.transform(self.Xi) if len(self.Xi) > 0 else None
                points = sb_sequence(
                    n=len(self.Xi) + 1,
                    d=self.space.transformed_n_dims,
                    existing_points=existing,
                    random_state=self._init_rng.randint(2**31),
                )
                return self.space.inverse_transform(np.atleast_2d(points[len(self.Xi)]))[0]
            return self.space.rvs()[0]
        if not self.gp.kernel_:
            raise RuntimeError("Initialization is finished, but no model has been fit.")
        return self._next_x

    def tell(
        self,
        x,
        y,
        noise_vector=None,
        fit=True,
        replace=False,
        n_samples=0,
        gp_samples=100,
        gp_burnin=10,
        progress=False,
    ):
        """Record observation(s), refit / resume the hyper-posterior MCMC and pick the next point
        (``bask/optimizer.py:228-380``; same arguments, same error behaviour)."""
        if replace:
            self.Xi = []
            self.yi = []
            self.noisei = []
            self._n_initial_points = self.n_initial_points_
        if is_listlike(y) and is_2Dlistlike(x):
            self.Xi.extend(x)
            self.yi.extend(y)
            if noise_vector is None:
                noise_vector = [0.0] * len(y)
            elif not is_listlike(noise_vector) or len(noise_vector) != len(y):
                raise ValueError("Vector of noise variances needs to be of equal length as `y`.")
            self.noisei.extend(noise_vector)
            self._n_initial_points -= len(y)
        elif is_listlike(x):
            self.Xi.append(x)
            self.yi.append(y)
            if noise_vector is None:
                noise_vector = 0.0
            elif is_listlike(noise_vector):
                raise ValueError("Vector of noise variances is a list, while tell only received one datapoint.")
            self.noisei.append(noise_vector)
            self._n_initial_points -= 1
        else:
            raise ValueError(f"Type of arguments `x` ({type(x)}) and `y` ({type(y)}) not compatible.")

        if fit and self._n_initial_points <= 0:
            if self.gp_priors is not None and len(self.gp_priors) != self.space.transformed_n_dims + 2:
                raise ValueError("The number of priors does not match the number of dimensions + 2.")
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                common = dict(
                    noise_vector=np.array(self.noisei),
                    priors=self.gp_priors,
                    n_desired_samples=gp_samples,
                    n_burnin=gp_burnin,
                    progress=progress,
                )
                if self.gp.pos_ is None or replace:
                    self.gp.fit(self.space.transform(self.Xi), self.yi, **common)
                else:
                    self.gp.sample(self.space.transform(self.Xi), self.yi, **common)

            if self.gp.warp_inputs:  # uniform in the WARPED space (bask/optimizer.py:353-357)
                X_warped = self.rng.uniform(size=(self.n_points, self.space.transformed_n_dims))
                X = self.gp.unwarp(X_warped)
            else:
                X = self.space.rvs_transformed(n_samples=self.n_points, random_state=self.rng)
            acq_values = evaluate_acquisitions(
                X=X,
                gpr=self.gp,
                acquisition_functions=(self.acq_func,),
                n_samples=n_samples,
                progress=False,
                random_state=self.rng.randint(0, np.iinfo(np.int32).max),
                **self.acq_func_kwargs,
            ).flatten()
            self._next_x = self.space.inverse_transform(X[np.argmax(acq_values)].reshape((1, -1)))[0]

        return create_result(self.Xi, self.yi, self.space, self.rng, models=[self.gp])

    def run(self, func, n_iter=1, replace=False, n_samples=5, gp_samples=100, gp_burnin=10):
        """ask/tell loop on an objective returning a value or a (value, noise variance) pair
        (``bask/optimizer.py:382-445``)."""
        for _ in range(n_iter):
            x = self.ask()
            out = func(x)
            if hasattr(out, "__len__"):
                val, noise = out
            else:
                val, noise = out, 0.0
            self.tell(x, val, noise_vector=noise, n_samples=n_samples, gp_samples=gp_samples, gp_burnin=gp_burnin,
                      replace=replace)
            replace = False
        return create_result(self.Xi, self.yi, self.space, self.rng, models=[self.gp])

    def _optimum_vs_space_draws(self, n_space_samples, n_gp_samples, n_random_starts, use_mean_gp, seed):
        """Function draws (device ``sample_y``) at [expected optimum, n_space_samples random points]:
        (1 + n_space_samples, n_gp_samples); row 0 belongs to the minimiser of the surrogate mean found by
        ``utils.expected_minimum`` (``bask/optimizer.py:493-512``).  ``seed`` is handed unchanged to each of the
        three consumers, as the reference does."""
        res = create_result(self.Xi, self.yi, self.space, self.rng, models=[self.gp])
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
