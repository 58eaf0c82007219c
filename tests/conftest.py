import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# small-n dpotrf with many OpenBLAS threads is pathological in this image
os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth(n, d, seed):
    """SURVEY.md 8(d) synthetic inputs (same generator as tests/golden/gen_golden.py)."""
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    return X, y


@pytest.fixture(scope="session")
def golden():
    return load_golden


def assert_variance_close(var_dev, var_ref, selfdiff, rtol=1e-6, factor=4.0):
    """The north star's tolerance on a predictive variance: 1e-6 RELATIVE -- except where the variance is the remainder of a
    cancellation ``k_** - k_*^T K^-1 k_*`` that the reference itself only knows to ``selfdiff``, the difference between its own
    two formulas (skopt's einsum with the explicit inverse vs scikit-learn's triangular solve; stored in the golden file or
    computed by ``oracle.predict_variance_selfdiff``).  There the bound is ``factor`` times the reference's own uncertainty,
    not a hand-picked number."""
    var_dev, var_ref = np.asarray(var_dev), np.asarray(var_ref)
    tol = np.maximum(rtol * np.abs(var_ref), factor * float(selfdiff))
    err = np.abs(var_dev - var_ref)
    worst = int(np.argmax(err - tol))
    assert np.all(err <= tol), ("variance off by %.3e at %d (value %.3e, 1e-6 relative = %.3e, %g x reference self-difference "
                                "= %.3e)" % (err[worst], worst, var_ref[worst], rtol * abs(var_ref[worst]), factor,
                                             factor * float(selfdiff)))
