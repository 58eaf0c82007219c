"""Initial-design sequences for the first ``n_initial_points`` asks (``bask/init.py``).

Outside the accelerated hot path (tiny, host-only); provided so that
``Optimizer(init_strategy="r2" | "sb")`` keeps working.
"""
import numpy as np
from scipy.optimize import minimize
from sklearn.utils import check_random_state

__all__ = ["r2_sequence", "sb_sequence"]

_PHI = {1: 1.61803398874989484820458683436563, 2: 1.32471795724474602596090885447809}


def _generalised_golden_ratio(d, n_iter=10):
    """Positive root of x^(d+1) = x + 1 (closed constants for d = 1, 2; fixed-point iteration
    otherwise, ``bask/init.py:90-98``)."""
    if d in _PHI:
        return _PHI[d]
    x = 2.0
    for _ in range(n_iter):
        x = (1.0 + x) ** (1.0 / (d + 1))
    return x


def phi(d, n_iter=10):
    """The reference's public name of the generalised golden ratio (``bask/init.py:90``)."""
    return _generalised_golden_ratio(d, n_iter)


def r2_sequence(n, d, seed=0.5):
    """First n points of the R_d additive-recurrence quasi-random sequence
    (``bask/init.py:101-128``): z_i = (seed + i * alpha) mod 1, alpha_j = phi_d^-(j+1) mod 1."""
    g = _generalised_golden_ratio(d)
    alpha = np.array([(1.0 / g) ** (j + 1) % 1 for j in range(d)])
    idx = np.arange(1, n + 1, dtype=np.float64)[:, None]
    return (seed + alpha[None, :] * idx) % 1


def _sb_energy(x, pts):
    """Steinerberger energy of a candidate x against the existing points
    sum_i prod_k (1 - log(2 sin(pi |x_k - p_ik|))); +inf where the log diverges
    (``bask/init.py:8-23``)."""
    diff = np.abs(np.asarray(x)[None, :] - pts)
    s = 2.0 * np.sin(np.pi * diff)
    if np.any(s == 0.0):
        return np.inf
    with np.errstate(invalid="ignore"):
        return float(np.sum(np.prod(1.0 - np.log(s), axis=1)))


def sb_sequence(n, d, existing_points=None, random_state=None, restarts=20):
    """Greedy Steinerberger low-discrepancy sequence in [0, 1]^d: each new point minimises the
    energy against all previous ones, global search by bounded quasi-Newton runs from ``restarts``
    uniform starts (``bask/init.py:26-89``; same arguments, same RNG consumption order)."""
    rng = check_random_state(random_state)
    if existing_points is None:
        pts = [rng.uniform(size=d)]
    else:
        pts = [np.asarray(p, dtype=np.float64) for p in existing_points]
        if len(pts) >= n:
            raise ValueError("No more points left to generate.")
    for _ in range(n - len(pts)):
        starts = rng.uniform(size=(restarts, d))
        arr = np.asarray(pts)
        best, best_val = starts[0], np.inf
        for x0 in starts:
            with np.errstate(invalid="ignore"):
                res = minimize(_sb_energy, x0=x0, bounds=[(0.0, 1.0)] * d, args=(arr,))
            if res.fun < best_val:
                best, best_val = res.x, res.fun
        pts.append(best)
    return np.asarray(pts)
