"""``BayesSearchCV``: cross-validated hyper-parameter search driven by the fully Bayesian ``Optimizer``
(host mirror of ``bask/searchcv.py``).

The reference subclasses ``skopt.BayesSearchCV`` and swaps in its own optimizer (``bask/searchcv.py:292-354``);
skopt is not part of this image, so the thin layer skopt adds on scikit-learn's ``BaseSearchCV`` is restated
here: the search spaces are normalised to ``(dict, n_iter)`` pairs, one ``Optimizer`` is made per space
(dimensions in sorted key order, ``acq_func="pvrs"`` unless given), and every iteration asks one point, lets
scikit-learn cross-validate it (``evaluate_candidates``) and tells the optimizer the negative mean test score
-- which is where the device hot path runs (``Optimizer.tell`` -> BayesGPR MCMC + acquisition)."""
import numpy as np
from sklearn.model_selection._search import BaseSearchCV
from sklearn.utils import check_random_state

from .optimizer import Optimizer
from .space import Dimension, _check_dimension

__all__ = ["BayesSearchCV"]


def dimensions_aslist(search_space):
    """Dimensions of a ``{name: dimension}`` space in sorted key order (skopt.utils.dimensions_aslist)."""
    return [search_space[k] for k in sorted(search_space.keys())]


def point_asdict(search_space, point_as_list):
    """``{name: value}`` for a point given in sorted key order (skopt.utils.point_asdict)."""
    return {k: v for k, v in zip(sorted(search_space.keys()), point_as_list)}


class BayesSearchCV(BaseSearchCV):
    """Same constructor as ``bask/searchcv.py:245-290``.  ``search_spaces``: a dict ``{param: dimension}``, a
    list of such dicts, or a list of ``(dict, n_iter)`` pairs; a dimension is a ``space.Real`` / ``Integer`` /
    ``Categorical`` or anything ``Optimizer(dimensions=...)`` accepts.  ``n_points`` other than 1 and
    ``return_policy="best_mean"`` behave as in the reference (the former is refused by ``Optimizer.ask``, the
    latter is stored but the best observed setting is what ``best_params_`` reports)."""

    def __init__(self, estimator, search_spaces, optimizer_kwargs=None, n_iter=50, return_policy="best_setting",
                 scoring=None, fit_params=None, n_jobs=1, n_points=1, iid=True, refit=True, cv=None, verbose=0,
                 pre_dispatch="2*n_jobs", random_state=None, error_score="raise", return_train_score=False):
        super().__init__(estimator=estimator, scoring=scoring, n_jobs=n_jobs, refit=refit, cv=cv, verbose=verbose,
                         pre_dispatch=pre_dispatch, error_score=error_score, return_train_score=return_train_score)
        self.search_spaces = search_spaces
        self.optimizer_kwargs = optimizer_kwargs
        self.n_iter = n_iter
        self.return_policy = return_policy
        self.fit_params = fit_params
        self.n_points = n_points
        self.iid = iid
        self.random_state = random_state

    # ------------------------------------------------------------------ search-space plumbing
    def _spaces(self):
        spaces = self.search_spaces
        if isinstance(spaces, dict):
            spaces = [spaces]
        out = []
        for entry in spaces:
            if isinstance(entry, tuple):
                space, n_iter = entry
                if not (isinstance(n_iter, (int, np.integer)) and n_iter > 0):
                    raise ValueError(f"Number of iterations in search space should be a positive integer, got {n_iter}")
            else:
                space, n_iter = entry, self.n_iter
            if not isinstance(space, dict):
                raise TypeError(f"Search space should be provided as a dict or list of dict, got {space}")
            for k, v in space.items():
                _check_dimension(v)  # raises on an invalid dimension
            out.append((space, int(n_iter)))
        return out

    @property
    def total_iterations(self):
        """Number of evaluations of the whole search (skopt's ``BayesSearchCV.total_iterations``)."""
        return sum(n for _, n in self._spaces())

    def _make_optimizer(self, params_space):
        """``bask/searchcv.py:292-318``."""
        kwargs = dict(self.optimizer_kwargs_)
        for k in ("n_samples", "gp_samples", "gp_burnin"):  # tell() arguments carried in optimizer_kwargs
            kwargs.pop(k, None)
        kwargs["dimensions"] = dimensions_aslist(params_space)
        optimizer = Optimizer(**kwargs)
        names = sorted(params_space.keys())
        for i, dim in enumerate(optimizer.space.dimensions):
            if isinstance(dim, Dimension) and dim.name is None:
                dim.name = names[i]
        return optimizer

    def _step(self, search_space, optimizer, evaluate_candidates, n_points=1):
        """One ask / cross-validate / tell round (``bask/searchcv.py:320-354``)."""
        params = [optimizer.ask(n_points=n_points)]
        params = [[np.array(v).item() for v in p] for p in params]
        params_dict = [point_asdict(search_space, p) for p in params]
        all_results = evaluate_candidates(params_dict)
        local_results = all_results["mean_test_score"][-len(params):]
        return optimizer.tell(params, [-float(score) for score in local_results], n_samples=self.n_samples_,
                              gp_samples=self.gp_samples_, gp_burnin=self.gp_burnin_, progress=False)

    # ------------------------------------------------------------------ BaseSearchCV hook
    def _run_search(self, evaluate_candidates):
        okw = dict(self.optimizer_kwargs or {})
        self.n_samples_ = okw.get("n_samples", 0)
        self.gp_samples_ = okw.get("gp_samples", 100)
        self.gp_burnin_ = okw.get("gp_burnin", 5)
        okw.setdefault("acq_func", "pvrs")
        rng = check_random_state(self.random_state)
        okw["random_state"] = rng.randint(0, np.iinfo(np.int32).max)
        self.optimizer_kwargs_ = okw
        self.optimizer_results_ = []
        for search_space, n_iter in self._spaces():
            optimizer = self._make_optimizer(search_space)
            result = None
            while n_iter > 0:
                n_points_adjusted = min(n_iter, self.n_points)
                result = self._step(search_space, optimizer, evaluate_candidates, n_points=n_points_adjusted)
                n_iter -= n_points_adjusted
            self.optimizer_results_.append(result)

    def fit(self, X, y=None, **params):
        """Run the search (and refit on the best setting).  ``fit_params`` given to the constructor are
        forwarded to the estimator's ``fit`` as in the reference."""
        merged = dict(self.fit_params or {})
        merged.update(params)
        return super().fit(X, y, **merged)
