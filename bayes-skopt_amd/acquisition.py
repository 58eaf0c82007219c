"""Acquisition functions and their evaluation driver (host side of SURVEY.md 8a row a9).

Mirrors ``bask/acquisition.py``: the same class hierarchy (``UncertaintyAcquisition`` /
``SampleAcquisition`` / ``FullGPAcquisition``), the same eight criteria and the
``evaluate_acquisitions`` driver.  What moved to the MI355X:

* ``evaluate_acquisitions`` builds the posteriors of ALL ``n_samples`` hyper-posterior draws in one
  batched device call and predicts mean/std at the candidates for all of them in one more
  (the reference rebuilds ``gpr.theta = chain_[i]`` + ``predict`` one draw at a time,
  ``bask/acquisition.py:112-141``);
* ``PVRS`` / ``VarianceReduction`` replace the per-candidate (n+1)x(n+1) Cholesky loop
  (``:287-300,328-338``) by the bordered-inverse identity evaluated with tile GEMMs on the device
  (``bgp_pvrs``) -- same numbers to rounding;
* the closed forms on (mu, std) (EI, TopTwoEI, LCB, mean, MES) are elementwise numpy on the host.
"""
from abc import ABC, abstractmethod

import numpy as np
import scipy.stats as st
from scipy.optimize import brentq
from scipy.special import ndtr
from sklearn.utils import check_random_state

__all__ = [
    "evaluate_acquisitions",
    "ExpectedImprovement",
    "TopTwoEI",
    "Expectation",
    "LCB",
    "MaxValueSearch",
    "ThompsonSampling",
    "VarianceReduction",
    "PVRS",
]


class Acquisition(ABC):
    @abstractmethod
    def __call__(self, *args, **kwargs):
        pass


class UncertaintyAcquisition(Acquisition, ABC):
    """Criteria computed from the predictive mean and standard deviation."""

    @abstractmethod
    def __call__(self, mu, std, *args, **kwargs):
        pass


class SampleAcquisition(Acquisition, ABC):
    """Criteria computed from one function realisation of the GP."""

    @abstractmethod
    def __call__(self, gp_sample, *args, **kwargs):
        pass


class FullGPAcquisition(Acquisition, ABC):
    """Criteria that need the whole (median) GP."""

    @abstractmethod
    def __call__(self, X, gp, *args, **kwargs):
        pass


# Closed forms with a device implementation (include/bgp.h BGP_ACQ_*).  Set to False to keep the means / standard
# deviations of all draws on the host and call the acquisition objects one draw at a time (tests compare both).
DEVICE_ACQUISITIONS = True
_ACQ_EI, _ACQ_MEAN, _ACQ_LCB, _ACQ_STD = 0, 1, 2, 3
_ACQ_MAX = 8


def _device_acq_spec(acq, kwargs):
    """(kind, parameter) of an acquisition the device evaluates itself, else None.  Exact types only: a subclass
    may redefine ``__call__``."""
    if type(acq) is ExpectedImprovement:
        y_opt = kwargs.get("y_opt")
        return _ACQ_EI, (np.nan if y_opt is None else float(y_opt))
    if type(acq) is Expectation:
        return _ACQ_MEAN, 0.0
    if type(acq) is LCB:
        alpha = kwargs.get("alpha", 1.96)
        return (_ACQ_STD, 0.0) if isinstance(alpha, str) and alpha == "inf" else (_ACQ_LCB, float(alpha))
    return None


def evaluate_acquisitions(X, gpr, acquisition_functions=None, n_samples=10, progress=False, random_state=None,
                          **kwargs):
    """Evaluate a set of acquisition functions on candidate points X (m, d).

    Same arguments, RNG consumption and averaging as ``bask/acquisition.py:48-147``:
    ``FullGPAcquisition``s are called once with the median GP; ``UncertaintyAcquisition``s /
    ``SampleAcquisition``s are averaged over ``n_samples`` chain rows drawn WITHOUT replacement, with
    the noise switched off (``noise_set_to_zero``); an output that is not all finite contributes
    zeros; ``gpr.theta`` is restored at the end.
    Returns (len(acquisition_functions), m).
    """
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    n_cand = len(X)
    acqs = list(acquisition_functions)
    out = np.zeros((len(acqs), n_cand))
    random_state = check_random_state(random_state)
    trace_i = random_state.choice(len(gpr.chain_), replace=False, size=n_samples)
    theta_backup = np.copy(gpr.theta)

    for i_acq, acq in enumerate(acqs):
        if isinstance(acq, FullGPAcquisition):
            vals = acq(X, gpr, random_state=random_state, **kwargs)
            if np.all(np.isfinite(vals)):
                out[i_acq] = vals

    has_unc = any(isinstance(a, UncertaintyAcquisition) for a in acqs)
    has_smp = any(isinstance(a, SampleAcquisition) for a in acqs)
    rows = gpr.chain_[trace_i]
    on_device = {}
    if len(trace_i) > 0 and has_unc:
        specs = [(j, _device_acq_spec(a, kwargs)) for j, a in enumerate(acqs) if isinstance(a, UncertaintyAcquisition)]
        if (DEVICE_ACQUISITIONS and not getattr(gpr, "warp_inputs", False) and hasattr(gpr, "_acq_hyper_samples")
                and not getattr(gpr, "_generic", False)
                and len(specs) <= _ACQ_MAX and all(sp is not None for _, sp in specs)):
            # build, predict, acquisition closed forms and the average over the draws in one device pass: the
            # (draws x candidates) means and variances stay in HBM
            vals = gpr._acq_hyper_samples(rows, X, [sp[0] for _, sp in specs], [sp[1] for _, sp in specs], n_samples)
            on_device = {j: vals[k] for k, (j, _) in enumerate(specs)}
        else:
            # ONE batched posterior build + ONE batched predict for all hyper-posterior draws
            mus, stds = gpr._predict_hyper_samples(rows, X, noise_zero=True)
    samples = None
    if len(trace_i) > 0 and has_smp:
        # one function realisation per draw, each from a chain row chosen by sample_y itself (bask/acquisition.py:132-136
        # -> bask/bayesgpr.py:679), with that row's kernel parameters and -- with input warping -- its own warp
        samples = gpr._sample_hyper_rows(len(trace_i), X, random_state)
    for j, vals in on_device.items():
        out[j] = vals
    for pos in range(len(trace_i)):
        for j, acq in enumerate(acqs):
            if j in on_device:
                continue
            if isinstance(acq, UncertaintyAcquisition):
                tmp = acq(mus[pos], stds[pos], **kwargs)
            elif isinstance(acq, SampleAcquisition):
                tmp = acq(samples[pos], **kwargs)
            else:
                continue
            if np.all(np.isfinite(tmp)):
                out[j] += tmp / n_samples
    if not np.array_equal(gpr.theta, theta_backup):
        gpr.theta = theta_backup
    return out


_SQRT_2PI = np.sqrt(2.0 * np.pi)


def _ei_f(x):
    # x Phi(x) + phi(x) with scipy's own kernels (norm.cdf is special.ndtr, norm.pdf is exp(-x^2/2)/sqrt(2 pi)): the
    # same values as ``st.norm.cdf(x)`` / ``st.norm.pdf(x)`` without the per-call overhead of the distribution
    # machinery (34 ms -> a few ms per tell at 128 hyper-samples x 10 000 candidates)
    return x * ndtr(x) + np.exp(-(x**2) / 2.0) / _SQRT_2PI


class ExpectedImprovement(UncertaintyAcquisition):
    """Expected improvement over the current optimum ``y_opt`` (default: min of mu)
    (``bask/acquisition.py:154-172``)."""

    def __call__(self, mu, std, *args, y_opt=None, **kwargs):
        if y_opt is None:
            y_opt = mu.min()
        values = np.zeros_like(mu)
        ok = std > 0
        values[ok] = _ei_f((y_opt - mu[ok]) / std[ok]) * std[ok]
        return values


class TopTwoEI(ExpectedImprovement):
    """Expected improvement over the point with the highest EI (``bask/acquisition.py:175-194``)."""

    def __call__(self, mu, std, *args, y_opt=None, **kwargs):
        ei = super().__call__(mu, std, *args, y_opt=y_opt, **kwargs)
        values = np.zeros_like(mu)
        top = np.argmax(ei)
        ok = std > 0
        spread = np.sqrt(std[ok] ** 2 + std[top] ** 2)
        values[ok] = spread * _ei_f((mu[top] - mu[ok]) / spread)
        return values


class Expectation(UncertaintyAcquisition):
    """Lowest predicted mean (``bask/acquisition.py:197-201``)."""

    def __call__(self, mu, std, *args, **kwargs):
        return -mu


class LCB(UncertaintyAcquisition):
    """Lower confidence bound ``alpha * std - mu``; ``alpha="inf"`` returns std
    (``bask/acquisition.py:204-216``)."""

    def __call__(self, mu, std, *args, alpha=1.96, **kwargs):
        if alpha == "inf":
            return std
        return alpha * std - mu


class MaxValueSearch(UncertaintyAcquisition):
    """Max-value entropy search (Wang & Jegelka 2017) with a Gumbel fit of the optimum distribution
    from three quantiles found by bisection (``bask/acquisition.py:219-267``; uses the global numpy
    RNG like the reference)."""

    def __call__(self, mu, std, *args, n_min_samples=1000, **kwargs):
        mean = -mu  # the algorithm is stated for maximisation

        def prob_below(x):
            return np.exp(np.sum(st.norm.logcdf((x - mean) / std), axis=0))

        lo = np.min(mean - 3 * std)
        hi = np.max(mean + 5 * std)
        q1, med, q2 = [brentq(lambda x, v=v: prob_below(x) - v, lo, hi) for v in (0.25, 0.5, 0.75)]
        beta = (q1 - q2) / (np.log(np.log(4.0 / 3.0)) - np.log(np.log(4.0)))
        alpha = med + beta * np.log(np.log(2.0))
        max_values = -np.log(-np.log(np.random.rand(n_min_samples).astype(np.float32))) * beta + alpha
        gamma = (max_values[None, :] - mean[:, None]) / std[:, None]
        norm = st.norm()
        return np.sum(gamma * norm.pdf(gamma) / (2.0 * norm.cdf(gamma)) - norm.logcdf(gamma), axis=1) / n_min_samples


class ThompsonSampling(SampleAcquisition):
    """Optimum of one sampled function (``bask/acquisition.py:270-274``)."""

    def __call__(self, gp_sample, *args, **kwargs):
        return -gp_sample


def _device_pvrs(gp, X, thompson_points):
    """covs[i] = trace(K_trans K_aug,i^-1 K_trans^T) for every candidate i on the device.  The
    augmented kernel matrix uses kernel_ as it stands (median noise included) and adds ``alpha`` only
    when it is a vector (``bask/acquisition.py:293-294,332-333``)."""
    has_vec = bool(np.iterable(gp.alpha))
    return gp._pvrs(X, thompson_points, has_vec)


class VarianceReduction(FullGPAcquisition):
    """Global variance reduction (``bask/acquisition.py:277-300``): PVRS with every candidate as a
    reference point."""

    def __call__(self, X, gp, *args, **kwargs):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        return _device_pvrs(gp, X, X)


class PVRS(FullGPAcquisition):
    """Predictive variance reduction search (Nguyen et al. 2017; ``bask/acquisition.py:303-339``):
    draw ``n_thompson`` functions from the median GP, take their minimisers among the candidates and
    score each candidate by how much observing it reduces the predictive variance at those points."""

    def __call__(self, X, gp, *args, n_thompson=10, random_state=None, **kwargs):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        thompson_sample = gp.sample_y(X, sample_mean=True, n_samples=n_thompson, random_state=random_state)
        thompson_points = X[np.argmin(thompson_sample, axis=0)]
        return _device_pvrs(gp, X, thompson_points)
