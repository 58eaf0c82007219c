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
from .bayesgpr import BayesGPR
from .optimizer import Optimizer
from .searchcv import BayesSearchCV
from .utils import construct_default_kernel, geometric_median, guess_priors, r2_sequence  # noqa: F401

__all__ = [
    "BayesGPR",
    "BayesSearchCV",
    "Optimizer",
    "acquisition",
    "geometric_median",
    "guess_priors",
    "construct_default_kernel",
    "r2_sequence",
]
