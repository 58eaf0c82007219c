"""Host helpers mirroring ``bask/utils.py``: chain summary (geometric median), default priors,
default kernel, input validation."""
import collections.abc

import numpy as np

from .init import r2_sequence  # noqa: F401  (re-exported like bask/utils.py:8-9)
from .kernels import ConstantKernel, Matern
from .priors import halfnorm_logpdf_logspace, make_roundflat

__all__ = ["geometric_median", "guess_priors", "construct_default_kernel", "validate_zeroone", "r2_sequence"]


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
