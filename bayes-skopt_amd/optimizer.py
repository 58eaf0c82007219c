"""Stepwise Bayesian optimisation: ``Optimizer.ask / tell / run`` (host mirror of
``bask/optimizer.py:35-445``).  ``tell`` is the per-iteration entry of the hot path: it
(re)fits / resumes the BayesGPR hyper-posterior MCMC on the device and evaluates the acquisition
function over ``n_points`` random candidates with the device predict / PVRS kernels.

Out of scope in this build: the arviz-based diagnostics
``probability_of_optimality`` / ``expected_optimality_gap`` / ``optimum_intervals``.
"""
import warnings

import numpy as np
from sklearn.utils import check_random_state

from . import acquisition
from .acquisition import evaluate_acquisitions
from .bayesgpr import BayesGPR
from .init import r2_sequence, sb_sequence
from .space import create_result, is_2Dlistlike, is_listlike, normalize_dimensions
from .utils import construct_default_kernel

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

    def probability_of_optimality(self, *args, **kwargs):
        raise NotImplementedError("post-hoc diagnostics (bask/optimizer.py:447-525) are outside the accelerated hot path")

    def expected_optimality_gap(self, *args, **kwargs):
        raise NotImplementedError("post-hoc diagnostics (bask/optimizer.py:527-620) are outside the accelerated hot path")

    def optimum_intervals(self, *args, **kwargs):
        raise NotImplementedError("arviz-based HDI diagnostics (bask/optimizer.py:622-689) are outside the accelerated hot path")
