"""MI355X-native implementation of the per-tell() hot path of kiudee/bayes-skopt (bask 0.11.0):
BayesGPR posterior (kernel-matrix build, jittered fp64 Cholesky, solves, log-marginal likelihood)
and the ensemble MCMC over kernel hyper-parameters, behind the ``bask`` API surface
(``bask/__init__.py:19-35`` export list).  Device code: hand-written HIP for gfx950 behind the
C-ABI of ``include/bgp.h`` (``lib/libbgp.so``), bound with ctypes.  No CPU fallback.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
