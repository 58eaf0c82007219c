"""MI355X-native implementation of the per-tell() hot path of kiudee/bayes-skopt (bask 0.11.0):
BayesGPR posterior (kernel-matrix build, jittered fp64 Cholesky, solves, log-marginal likelihood)
and the ensemble MCMC over kernel hyper-parameters, behind the ``bask`` API surface
(``bask/__init__.py:19-35`` export list).  Device code: hand-written HIP for gfx950 behind the
C-ABI of ``include/bgp.h`` (``lib/libbgp.so``), bound with ctypes.  No CPU fallback.

    import bayes_skopt_amd as bask
    gp = bask.BayesGPR(kernel=...); gp.fit(X, y); gp.predict(Xq, return_std=True)
"""
__version__ = "0.1.0"

from . import _lib, acquisition, distributed, init, kernels, priors, sampler, searchcv, space, utils  # noqa: F401
from .acquisition import *  # noqa: F401,F403  (bask/__init__.py:12: the acquisition classes are top-level names)
from .bayesgpr import BayesGPR
from .init import r2_sequence, sb_sequence
from .optimizer import Optimizer
from .searchcv import BayesSearchCV
from .utils import construct_default_kernel, geometric_median, guess_priors  # noqa: F401

# bask/__init__.py:19-35, name for name, then this package's additions
__all__ = [
    "BayesGPR",
    "Optimizer",
    "BayesSearchCV",
    "guess_priors",
    "evaluate_acquisitions",
    "ExpectedImprovement",
    "TopTwoEI",
    "Expectation",
    "LCB",
    "MaxValueSearch",
    "r2_sequence",
    "sb_sequence",
    "ThompsonSampling",
    "VarianceReduction",
    "PVRS",
    "acquisition",
    "geometric_median",
    "construct_default_kernel",
]
