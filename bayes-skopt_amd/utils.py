"""Host helpers mirroring ``bask/utils.py``: chain summary (geometric median), default priors,
default kernel, input validation."""
import collections.abc
import math

import numpy as np

from .init import r2_sequence  # noqa: F401  (re-exported like bask/utils.py:8-9)
from .kernels import ConstantKernel, Matern
from .priors import halfnorm_logpdf_logspace, make_roundflat

__all__ = ["expected_minimum", "hdi", "geometric_median", "guess_priors", "construct_default_kernel", "validate_zeroone", "r2_sequence",
           "get_progress_bar"]


def geometric_median(X, eps=1e-5):
    """Geometric median (point minimising the summed Euclidean distance) of the rows of X.

    Weiszfeld's fixed-point iteration started at the mean, with the Vardi-Zhang correction when
    the iterate coincides with data rows, stopped when two iterates are closer than ``eps`` --
    the same algorithm and stopping rule as ``bask/utils.py:21-65``.
    """
    X = np.asarray(X, dtype=np.float64)
    n_pts = X.shape[0]
    cur = X.mean(axis=0)
    while True:
        dist = np.sqrt(np.einsum("ij,ij->i", X - cur, X - cur))
        away = dist != 0
        n_at = n_pts - int(np.count_nonzero(away))
        if n_at == n_pts:
            return cur
        w = 1.0 / dist[away]
        w_sum = w.sum()
        target = (w[:, None] / w_sum * X[away]).sum(axis=0)
        if n_at == 0:
            nxt = target
        else:
            pull = (target - cur) * w_sum
            r = np.linalg.norm(pull)
            shrink = 0.0 if r == 0 else n_at / r
            nxt = max(0.0, 1.0 - shrink) * target + min(1.0, shrink) * cur
        if np.sqrt(((cur - nxt) ** 2).sum()) < eps:
            return nxt
        cur = nxt


def _collect_priors(kernel, out):
    if hasattr(kernel, "kernel"):  # unary wrappers (Exponentiation)
        _collect_priors(kernel.kernel, out)
    elif hasattr(kernel, "k1"):  # Sum / Product
        _collect_priors(kernel.k1, out)
        _collect_priors(kernel.k2, out)
    elif hasattr(kernel, "kernels"):  # CompoundKernel
        for k in kernel.kernels:
            _collect_priors(k, out)
    else:
        name = type(kernel).__name__
        if name == "ConstantKernel":
            if kernel.constant_value_bounds == "fixed":
                return
            out.append(halfnorm_logpdf_logspace(2.0))
        elif name == "WhiteKernel":
            if kernel.noise_level_bounds == "fixed":
                return
            out.append(halfnorm_logpdf_logspace(2.0))
        elif name in ("Matern", "RBF"):
            if isinstance(kernel.length_scale, (collections.abc.Sequence, np.ndarray)):
                count = len(kernel.length_scale)
            else:
                count = 1
            roundflat = make_roundflat(lower_bound=0.1, upper_bound=0.6, lower_steepness=2.0, upper_steepness=8.0)

            def ls_prior(t, _rf=roundflat):
                t = np.asarray(t, dtype=np.float64)
                with np.errstate(over="ignore"):
                    out_ = _rf(np.exp(t)) + t
                return float(out_) if np.ndim(out_) == 0 else out_

            _lo, _hi, _plo, _phi, _ln = roundflat._bgp_roundflat
            ls_prior._bgp_device = (2, (math.log(_lo), math.log(_hi), _plo, _phi, _ln))  # include/bgp.h bgp_mcmc_run, prior_kind 2
            out.extend([ls_prior] * count)
        else:
            raise NotImplementedError(f"Unable to guess priors for this kernel: {kernel}.")


def guess_priors(kernel):
    """One log-prior callable per entry of ``kernel.theta`` (same order): half-Normal(0, 2) on the
    square root of every signal variance / noise level, round-flat(0.1, 0.6) on every length scale,
    both with the log-space change of variables (``bask/utils.py:68-124,154-179``)."""
    priors = []
    _collect_priors(kernel, priors)
    return priors


def construct_default_kernel(dimensions):
    """``ConstantKernel(1.0, (0.1, 2.0)) * Matern([0.3]*d, (0.2, 0.5), nu=2.5)``
    (``bask/utils.py:127-151``)."""
    d = len(dimensions)
    return ConstantKernel(constant_value=1.0, constant_value_bounds=(0.1, 2.0)) * Matern(
        length_scale=[0.3] * d, length_scale_bounds=(0.2, 0.5), nu=2.5
    )


def validate_zeroone(arr):
    """Raise ValueError unless every entry lies in [0, 1] (``bask/utils.py:212-228``)."""
    arr = np.asarray(arr)
    if np.any(arr < 0) or np.any(arr > 1):
        raise ValueError("Not all values of the array are between 0 and 1.")


def expected_minimum(res, n_random_starts=20, random_state=None):
    """Minimum of the surrogate's predictive mean: L-BFGS-B from the best observed point and
    ``n_random_starts`` random points of the space (restatement of ``skopt.utils.expected_minimum``, the
    routine ``bask/optimizer.py:497-503`` calls).  Returns (x in the original space, predicted value).

    skopt lets scipy difference the objective one point at a time; here every iterate and its 2-point
    difference stencil go to the device as ONE predict batch (d + 1 rows), forward differences switching
    to backward ones at the upper bound like scipy's ``approx_derivative``."""
    from scipy.optimize import minimize
    from sklearn.utils import check_random_state

    space = res.space
    if space.is_partly_categorical:
        raise ValueError("expected_minimum does not support any categorical values")
    reg = res.models[-1]
    bounds = np.asarray(space.bounds, dtype=np.float64)
    d = len(bounds)
    eps = np.sqrt(np.finfo(np.float64).eps)

    def fun_and_grad(x):
        h = eps * np.maximum(1.0, np.abs(x))
        sign = np.where(x + h > bounds[:, 1], -1.0, 1.0)
        pts = np.tile(x, (d + 1, 1))
        pts[1:, :][np.arange(d), np.arange(d)] += sign * h
        vals = np.asarray(reg.predict(space.transform(pts.tolist())), dtype=np.float64)
        return float(vals[0]), (vals[1:] - vals[0]) / (sign * h)

    rng = check_random_state(random_state)
    xs = [res.x]
    if n_random_starts > 0:
        xs.extend(space.rvs(n_random_starts, random_state=rng))
    best_x, best_fun = None, np.inf
    for x0 in xs:
        r = minimize(fun_and_grad, x0=np.asarray(x0, dtype=np.float64), jac=True, bounds=space.bounds, method="L-BFGS-B")
        if r.fun < best_fun:
            best_x, best_fun = r.x, r.fun
    return [float(v) for v in best_x], float(best_fun)


def hdi(samples, hdi_prob=0.95, multimodal=False, max_modes=10, grid=512):
    """Highest density interval(s) of a 1-d sample (the quantity ``bask/optimizer.py:684`` takes from
    ``arviz.hdi``; arviz is not part of this image, so its two estimators are restated):

    * ``multimodal=False``: the narrowest interval containing ``hdi_prob`` of the sorted sample -> (2,);
    * ``multimodal=True``: density estimate on a regular grid (Gaussian KDE, Silverman bandwidth), grid cells
      taken in order of decreasing density until they hold ``hdi_prob`` of the mass, contiguous runs of
      cells reported as separate intervals -> (n_modes, 2)."""
    x = np.sort(np.asarray(samples, dtype=np.float64).ravel())
    x = x[np.isfinite(x)]
    n = len(x)
    if n == 0:
        raise ValueError("hdi needs at least one finite sample")
    if not multimodal:
        inc = min(int(np.floor(hdi_prob * n)), n - 1)
        widths = x[inc:] - x[: n - inc]
        i = int(np.argmin(widths))
        return np.array([x[i], x[i + inc]])
    lower, upper = x[0], x[-1]
    if upper <= lower:
        return np.array([[lower, upper]])
    from scipy.stats import gaussian_kde

    bins = np.linspace(lower, upper, grid)
    density = gaussian_kde(x, bw_method="silverman")(bins)
    dx = (upper - lower) / grid
    density = density * dx
    density = density / density.sum()
    order = np.argsort(-density)
    keep = np.sort(bins[order][np.cumsum(density[order]) <= hdi_prob])
    if keep.size == 0:
        return np.array([[lower, upper]])
    step = bins[1] - bins[0]
    runs = np.split(keep, np.where(np.diff(keep) >= step * 1.1)[0] + 1)
    return np.array([[r[0], r[-1]] for r in runs[:max_modes]])


def get_progress_bar(display, total):
    """``bask/utils.py:198-209``: a tqdm bar of ``total`` ticks when ``display is True``, an object with the same ``update`` /
    ``close`` / context-manager interface that does nothing otherwise (or when tqdm cannot be imported)."""
    from .sampler import _progress

    return _progress(display is True, total)
