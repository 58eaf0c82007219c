"""Randomised parity sweep of the LML / gradient / posterior / predict entry points against the oracle over random
sizes (1 .. 900 points: every tile-edge case of the 128-blocked factorisation), dimensions, kernel families, forms,
batch sizes and hyper-parameters, including vector alpha and deliberately near-singular cases.

Tolerances: the north star's 1e-6 relative on LML, alpha, posterior mean and posterior variance whenever
cond(K) <= COND_BOUND (alpha / mean relative to their largest entry, the variance relative to the prior variance it
is the remainder of: ``var = k** - q`` is a difference of O(k**) numbers, so below that scale neither this path nor
LAPACK's has significant digits).  Above the bound both paths lose digits at the rate cond * eps; those draws are
checked at cond-scaled tolerances and their worst errors are printed (``pytest -s``)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STATS = ["rbf", "matern12", "matern32", "matern52"]
COND_BOUND = 1e7


@pytest.fixture(scope="module")
def lib():
    import bayes_skopt_amd  # noqa: F401
    from bayes_skopt_amd import _lib

    return _lib


@pytest.fixture(scope="module")
def O():
    from oracle import gp_oracle

    return gp_oracle


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("BGP_STRESS_N", "12"))))
def test_random_problem_matches_oracle(lib, O, seed):
    rng = np.random.RandomState(1000 + seed)
    n = int(rng.choice([1, 2, 15, 16, 17, 127, 128, 129, 255, 256, 257, 300, 511, 640, 777, 900])) if seed < 8 \
        else int(rng.randint(1, 900))
    d = int(rng.randint(1, 7))
    stat, form = STATS[rng.randint(4)], ("product", "sum")[rng.randint(2)]
    X = rng.uniform(size=(n, d))
    if n > 3 and seed % 3 == 0:
        X[1] = X[0] + 1e-9  # nearly duplicate inputs: conditioning decided by the noise / alpha
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    alpha = np.full(n, 1e-10) if seed % 2 == 0 else 10.0 ** rng.uniform(-8, -2, size=n)
    B = int(rng.randint(1, 9))
    H = np.column_stack([rng.uniform(-1.5, 1.0, B), rng.uniform(np.log(0.05), np.log(2.0), (B, d)),
                         rng.uniform(np.log(1e-4), np.log(0.5), B)])
    ctx = lib.Context(X, y, alpha, form=form, stationary=stat, max_batch=4)  # B > 4 exercises chunking
    got, status = ctx.lml(H, return_status=True)
    ref = O.lml_batch(X, y, alpha, H, stationary=stat, form=form)
    assert np.all(status == 0)
    np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-9)
    # gradient, posterior factors and predictions for the first two hyper-parameter vectors
    for h in H[:2]:
        l1, g1 = ctx.lml_grad(h[None, :])[:2]
        l0, g0 = O.lml_and_grad(X, y, alpha, h, stationary=stat, form=form)
        np.testing.assert_allclose(l1[0], l0, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(g1[0], g0, rtol=1e-5, atol=1e-6 * (1 + np.abs(g0).max()))
        cond = np.linalg.cond(O.gram_with_jitter(X, alpha, h, stationary=stat, form=form))
        # 1e-6 below the bound; beyond it the digits go at cond * eps on either path (factor 50: two n^3 passes)
        tol = 1e-6 if cond <= COND_BOUND else max(1e-6, 50 * cond * np.finfo(float).eps)
        res = ctx.posterior(h[None, :], want_L=True, want_alpha=True, want_K_inv=True)
        Lo, Ko, ao = O.posterior(X, y, alpha, h, stationary=stat, form=form)
        np.testing.assert_allclose(res["L"][0], Lo, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(res["alpha"][0], ao, rtol=tol, atol=tol * np.abs(ao).max())
        np.testing.assert_allclose(res["K_inv"][0], Ko, rtol=tol, atol=tol * np.abs(Ko).max())
        Xq = rng.uniform(size=(int(rng.randint(1, 200)), d))
        mean, var = ctx.predict(h[None, :], Xq)
        mo, so = O.predict(X, y, alpha, h, Xq, stationary=stat, form=form)
        prior_var = O.kernel_diag(1, h, d, form=form)[0]
        np.testing.assert_allclose(mean[0], mo, rtol=tol, atol=tol * (np.abs(mo).max() + 1e-300))
        np.testing.assert_allclose(var[0], so**2, rtol=tol, atol=tol * prior_var)
        if cond > COND_BOUND:
            print("cond %.2e (n=%d %s/%s): worst rel err alpha %.1e  mean %.1e  var/prior %.1e" % (
                cond, n, stat, form, np.abs(res["alpha"][0] - ao).max() / np.abs(ao).max(),
                np.abs(mean[0] - mo).max() / (np.abs(mo).max() + 1e-300), np.abs(var[0] - so**2).max() / prior_var))
    ctx.close()
