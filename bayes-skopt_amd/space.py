"""Minimal search-space layer for ``Optimizer`` (the subset of ``skopt.space`` / ``skopt.utils`` the
reference's ask/tell loop touches: ``normalize_dimensions``, ``Space.transform / inverse_transform /
rvs``, ``create_result``; ``bask/optimizer.py:7-13,144,353-380``).  skopt is absent from this image, so
this is written from its documented behaviour: every dimension is mapped to [0, 1] ("normalize"
transform); Integer dimensions round on the way back; Categorical dimensions are label-encoded to
equally spaced points in [0, 1].  Host-only bookkeeping -- nothing here is on the device hot path.
"""
import numbers

import numpy as np
from scipy.optimize import OptimizeResult
from sklearn.utils import check_random_state

__all__ = ["Real", "Integer", "Categorical", "Space", "normalize_dimensions", "create_result", "is_listlike",
           "is_2Dlistlike"]


def is_listlike(x):
    return isinstance(x, (list, tuple, np.ndarray))


def is_2Dlistlike(x):
    return is_listlike(x) and len(x) > 0 and all(is_listlike(xi) for xi in x)


class Dimension:
    transformed_size = 1
    name = None

    def rvs(self, n_samples=1, random_state=None):
        rng = check_random_state(random_state)
        return self.inverse_transform(rng.uniform(size=n_samples))


class Real(Dimension):
    def __init__(self, low, high, prior="uniform", name=None):
        if high <= low:
            raise ValueError(f"the lower bound {low} has to be less than the upper bound {high}")
        if prior not in ("uniform", "log-uniform"):
            raise ValueError(f"prior should be 'uniform' or 'log-uniform', got {prior}")
        self.low, self.high, self.prior, self.name = float(low), float(high), prior, name

    @property
    def bounds(self):
        return (self.low, self.high)

    def transform(self, x):
        x = np.asarray(x, dtype=np.float64)
        if self.prior == "log-uniform":
            return (np.log10(x) - np.log10(self.low)) / (np.log10(self.high) - np.log10(self.low))
        return (x - self.low) / (self.high - self.low)

    def inverse_transform(self, xt):
        xt = np.asarray(xt, dtype=np.float64)
        if self.prior == "log-uniform":
            v = 10.0 ** (xt * (np.log10(self.high) - np.log10(self.low)) + np.log10(self.low))
        else:
            v = xt * (self.high - self.low) + self.low
        return np.clip(v, self.low, self.high)


class Integer(Dimension):
    def __init__(self, low, high, name=None):
        if high <= low:
            raise ValueError(f"the lower bound {low} has to be less than the upper bound {high}")
        self.low, self.high, self.name = int(low), int(high), name

    @property
    def bounds(self):
        return (self.low, self.high)

    def transform(self, x):
        return (np.asarray(x, dtype=np.float64) - self.low) / (self.high - self.low)

    def inverse_transform(self, xt):
        v = np.round(np.asarray(xt, dtype=np.float64) * (self.high - self.low) + self.low)
        return np.clip(v, self.low, self.high).astype(int)


class Categorical(Dimension):
    def __init__(self, categories, name=None):
        self.categories = list(categories)
        if len(self.categories) < 2:
            raise ValueError("a Categorical dimension needs at least two categories")
        self.name = name

    @property
    def bounds(self):
        return tuple(self.categories)

    def transform(self, x):
        idx = np.array([self.categories.index(v) for v in np.atleast_1d(np.asarray(x, dtype=object))], dtype=np.float64)
        return idx / (len(self.categories) - 1)

    def inverse_transform(self, xt):
        idx = np.clip(np.round(np.asarray(xt, dtype=np.float64) * (len(self.categories) - 1)), 0,
                      len(self.categories) - 1).astype(int)
        return np.array([self.categories[i] for i in np.atleast_1d(idx)], dtype=object)


def _check_dimension(dim):
    if isinstance(dim, Dimension):
        return dim
    if not is_listlike(dim):
        raise ValueError(f"Dimension has to be a list or tuple, got {dim!r}")
    if len(dim) == 3 and isinstance(dim[2], str) and all(isinstance(v, numbers.Real) for v in dim[:2]):
        return Real(dim[0], dim[1], prior=dim[2])
    if len(dim) == 2 and all(isinstance(v, numbers.Integral) and not isinstance(v, bool) for v in dim):
        return Integer(*dim)
    if len(dim) == 2 and all(isinstance(v, numbers.Real) and not isinstance(v, bool) for v in dim):
        return Real(*dim)
    return Categorical(dim)


class Space:
    def __init__(self, dimensions):
        self.dimensions = [_check_dimension(d) for d in dimensions]

    @property
    def n_dims(self):
        return len(self.dimensions)

    @property
    def transformed_n_dims(self):
        return sum(d.transformed_size for d in self.dimensions)

    @property
    def bounds(self):
        return [d.bounds for d in self.dimensions]

    @property
    def is_partly_categorical(self):
        return any(isinstance(d, Categorical) for d in self.dimensions)

    def _columns(self, X):
        if len(X) == 0:
            return [[] for _ in self.dimensions]
        rows = [list(x) for x in X]
        if any(len(r) != self.n_dims for r in rows):
            raise ValueError(f"every point needs {self.n_dims} coordinates")
        return [[r[j] for r in rows] for j in range(self.n_dims)]

    def transform(self, X):
        """list of points (original space) -> (n, d) array in [0, 1]^d."""
        cols = self._columns(X)
        return np.column_stack([np.asarray(d.transform(c), dtype=np.float64) for d, c in zip(self.dimensions, cols)])

    def inverse_transform(self, Xt):
        """(n, d) array in [0, 1]^d -> list of points in the original space."""
        Xt = np.atleast_2d(np.asarray(Xt, dtype=np.float64))
        cols = [d.inverse_transform(Xt[:, j]) for j, d in enumerate(self.dimensions)]
        return [[_py(c[i]) for c in cols] for i in range(Xt.shape[0])]

    def rvs(self, n_samples=1, random_state=None):
        rng = check_random_state(random_state)
        cols = [d.rvs(n_samples=n_samples, random_state=rng) for d in self.dimensions]
        return [[_py(c[i]) for c in cols] for i in range(n_samples)]

    def rvs_transformed(self, n_samples=1, random_state=None):
        """``transform(rvs(n_samples))`` as one (n_samples, d) array without the detour through Python
        lists: same RNG consumption and the same values (every dimension still goes through its
        inverse transform -- rounding for Integer, snapping for Categorical -- and back).  Used for the
        candidate grid of ``Optimizer.tell`` (``bask/optimizer.py:358-363``), where building 10 000 x d Python
        objects costs more than the device work."""
        rng = check_random_state(random_state)
        cols = []
        for dim in self.dimensions:
            vals = dim.rvs(n_samples=n_samples, random_state=rng)
            cols.append(np.asarray(dim.transform(vals), dtype=np.float64))
        return np.column_stack(cols)


def _py(v):
    return v.item() if isinstance(v, np.generic) else v


def normalize_dimensions(dimensions):
    """A Space whose transform maps every dimension to [0, 1] (all transforms here do)."""
    return dimensions if isinstance(dimensions, Space) else Space(dimensions)


def create_result(Xi, yi, space=None, rng=None, specs=None, models=None):
    """``scipy.optimize.OptimizeResult`` with the fields skopt's ``create_result`` fills
    (``bask/optimizer.py:378-380``)."""
    res = OptimizeResult()
    yi = np.asarray(yi)
    if len(yi) > 0:
        best = int(np.argmin(yi))
        res.x = Xi[best]
        res.fun = yi[best]
    else:
        res.x, res.fun = None, None
    res.func_vals = yi
    res.x_iters = Xi
    res.models = [] if models is None else models
    res.space = space
    res.random_state = rng
    res.specs = specs
    return res
