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
