"""Prior distributions over kernel hyper-parameters (host side, evaluated on every proposal).

Mirrors ``bask/priors.py`` (``make_roundflat``).  The returned callables accept scalars *or* numpy
arrays, so the ensemble sampler evaluates one prior per hyper-parameter column for a whole
half-step instead of one Python call per walker per dimension (the reference's dominant cost at
small n).
"""
import functools
import math

import numpy as np
from scipy.integrate import quad

__all__ = ["make_roundflat", "halfnorm_logpdf_logspace"]


def make_roundflat(
    lower_bound=0.1,
    upper_bound=0.6,
    lower_steepness=2.0,
    upper_steepness=8.0,
    integration_bounds=(0.0, 10.0),
):
    """Round-flat log-prior: roughly flat on (lower_bound, upper_bound), smooth power-law walls.

    Same parameters, same density and same normalisation (numerical integral of exp(shape) over
    ``integration_bounds``) as ``bask/priors.py:7-57``.
    """
    lo, hi = float(lower_bound), float(upper_bound)
    p_lo, p_hi = -2.0 * float(lower_steepness), 2.0 * float(upper_steepness)

    def shape(x):
        x = np.asarray(x, dtype=np.float64)
        with np.errstate(over="ignore", divide="ignore"):
            return -2.0 * ((x / lo) ** p_lo + (x / hi) ** p_hi)

    log_norm = _roundflat_log_norm(lo, hi, p_lo, p_hi, float(integration_bounds[0]), float(integration_bounds[1]))

    def prior(x):
        out = shape(x) - log_norm
        return float(out) if np.ndim(out) == 0 else out

    prior._bgp_roundflat = (lo, hi, p_lo, p_hi, log_norm)  # (the device-resident sampler evaluates the same expression)
    return prior


@functools.lru_cache(maxsize=64)
def _roundflat_log_norm(lo, hi, p_lo, p_hi, a, b):
    """log of the numerical integral of exp(shape) over (a, b) (``bask/priors.py:48-52``).  ``guess_priors`` builds
    the same default prior on every ``tell``: the quadrature (1.5 ms) is done once per parameter set."""
    def dens(t):
        with np.errstate(over="ignore", divide="ignore"):
            return math.exp(-2.0 * float(np.float64(t / lo) ** p_lo + np.float64(t / hi) ** p_hi))

    return math.log(quad(dens, a, b)[0])


_HALFNORM_CONST = 0.5 * math.log(2.0 / math.pi)


def halfnorm_logpdf_logspace(scale):
    """log-density of ``sqrt(exp(t))`` ~ HalfNormal(scale) expressed in t (log-variance) space:
    ``halfnorm(scale).logpdf(sqrt(exp(t))) + t/2 - log 2`` -- the prior ``bask/utils.py:95-99``
    puts on signal variance and noise."""
    scale = float(scale)
    c = _HALFNORM_CONST - math.log(scale) - math.log(2.0)

    def prior(t):
        t = np.asarray(t, dtype=np.float64)
        with np.errstate(over="ignore"):
            out = c - 0.5 * np.exp(t) / (scale * scale) + 0.5 * t
        return float(out) if np.ndim(out) == 0 else out

    prior._bgp_device = (1, (c, scale * scale, 0.0, 0.0, 0.0))  # include/bgp.h bgp_mcmc_run, prior_kind 1
    return prior
