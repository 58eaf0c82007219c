"""Input warping (warp_inputs=True, bask/bayesgpr.py:219-316,353-365): device Beta-CDF warp, per-walker
warped LML, context-level warp for posterior/predict, and the BayesGPR / Optimizer surface."""
import numpy as np
import pytest
from scipy.stats import beta as sbeta

from conftest import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import bayes_skopt_amd  # noqa: F401
    from bayes_skopt_amd import _lib

    assert _lib.device_count() >= 1
    return _lib


@pytest.fixture(scope="module")
def O():
    from oracle import gp_oracle

    return gp_oracle


def test_device_beta_cdf_matches_scipy(lib):
    n, d = 500, 4
    X = np.random.RandomState(0).uniform(size=(n, d))
    X[0] = 0.0
    X[1] = 1.0
    X[2] = 1e-12
    X[3] = 1 - 1e-12
    ctx = lib.Context(X, np.zeros(n), 1e-10, max_batch=2)
    for seed, scale in ((1, 0.3), (2, 1.0), (3, 2.0)):
        w = scale * np.random.RandomState(seed).randn(2 * d)
        got = ctx.beta_cdf(X, w)
        ref = np.column_stack([sbeta(np.exp(w[k]), np.exp(w[d + k])).cdf(X[:, k]) for k in range(d)])
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-15)
    ctx.close()


def test_per_walker_warped_lml(lib, O):
    n, d, B = 260, 3, 10
    X, y = synth(n, d, 12)
    rng = np.random.RandomState(13)
    H = np.array([0.0, -1.0, -1.1, -0.9, -3.5]) + 0.15 * rng.randn(B, d + 2)
    W = 0.3 * rng.randn(B, 2 * d)
    ctx = lib.Context(X, y, 1e-10, max_batch=4)  # chunked
    got = ctx.lml_warped(H, W)
    ref = np.array([O.lml_warped(X, y, np.full(n, 1e-10), H[b], W[b]) for b in range(B)])
    np.testing.assert_allclose(got, ref, rtol=1e-6)
    # zero warp parameters = Beta(1, 1) = identity
    np.testing.assert_allclose(ctx.lml_warped(H, np.zeros_like(W)), ctx.lml(H), rtol=1e-12)
    ctx.close()


def test_context_level_warp_posterior_predict(lib, O):
    n, d, m = 150, 2, 60
    X, y = synth(n, d, 21)
    Xq = np.random.RandomState(22).uniform(size=(m, d))
    th = np.array([0.1, -1.0, -1.2, -3.0])
    w = np.array([0.2, -0.3, 0.4, 0.1])
    ctx = lib.Context(X, y, 1e-10, max_batch=2)
    ctx.set_warp(w)
    np.testing.assert_allclose(ctx.lml(th)[0], O.lml_warped(X, y, np.full(n, 1e-10), th, w), rtol=1e-6)
    ctx.posterior(th)
    mean, var = ctx.predict(th, Xq)
    mo, so = O.predict(O.warp_inputs(X, w), y, np.full(n, 1e-10), th, O.warp_inputs(Xq, w))
    np.testing.assert_allclose(mean[0], mo, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(var[0]), so, rtol=1e-6, atol=1e-8)
    ctx.set_warp(None)
    np.testing.assert_allclose(ctx.lml(th)[0], O.lml(X, y, np.full(n, 1e-10), th), rtol=1e-6)
    ctx.close()


def test_bayesgpr_with_warping_end_to_end(O):
    """A function that is stationary only after a monotone warp of its input: the warped GP must run,
    keep the reference's chain layout (p + 2d columns) and predict consistently with the oracle."""
    import bayes_skopt_amd as bask

    rng = np.random.RandomState(0)
    n, d = 80, 1
    X = rng.uniform(size=(n, d))
    y = np.sin(12.0 * X[:, 0] ** 3) + 0.05 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel([0]), random_state=1, warp_inputs=True)
    gp.fit(X, y, n_desired_samples=60, n_burnin=5, n_walkers_per_thread=30, progress=False)
    assert gp.chain_.shape == (60, 3 + 2 * d)
    assert gp.warp_alphas_.shape == (d,) and gp.warp_betas_.shape == (d,)
    w = np.concatenate([gp.warp_alphas_, gp.warp_betas_])
    np.testing.assert_allclose(gp.X_train_, O.warp_inputs(X, w), rtol=1e-10, atol=1e-14)
    Xq = np.linspace(0.01, 0.99, 25)[:, None]
    mean, std = gp.predict(Xq, return_std=True)
    mo, so = O.predict(O.warp_inputs(X, w), y, np.full(n, 1e-10), gp.theta, O.warp_inputs(Xq, w))
    np.testing.assert_allclose(mean, mo, rtol=1e-6, atol=1e-8)
    from conftest import assert_variance_close

    assert_variance_close(std**2, so**2, O.predict_variance_selfdiff(O.warp_inputs(X, w), y, np.full(n, 1e-10), gp.theta,
                                                                      O.warp_inputs(Xq, w)))
    with pytest.raises(ValueError):
        gp.predict(np.array([[1.5]]))
    np.testing.assert_allclose(gp.unwarp(gp.warp(Xq)), Xq, rtol=1e-8)
    # acquisition driver with per-draw warps, then everything is restored
    acq = bask.acquisition.evaluate_acquisitions(Xq, gp, [bask.acquisition.ExpectedImprovement(),
                                                        bask.acquisition.PVRS()], n_samples=3, random_state=0)
    assert acq.shape == (2, 25) and np.all(np.isfinite(acq))
    np.testing.assert_allclose(np.concatenate([gp.warp_alphas_, gp.warp_betas_]), w)
    s = gp.sample_y(Xq, n_samples=2, random_state=1)
    assert s.shape == (25, 2) and np.all(np.isfinite(s))
    np.testing.assert_allclose(np.concatenate([gp.warp_alphas_, gp.warp_betas_]), w)


def test_optimizer_with_warping():
    import bayes_skopt_amd as bask

    rng = np.random.RandomState(0)
    opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * 2, n_points=200, n_initial_points=6, init_strategy="r2",
                         gp_kwargs=dict(warp_inputs=True), acq_func="pvrs", random_state=0)
    for _ in range(8):
        x = opt.ask()
        opt.tell(x, float(np.sin(5 * x[0] ** 2) + x[1] + 0.01 * rng.randn()), gp_samples=100, gp_burnin=2)
    assert opt.gp.chain_.shape == (100, 4 + 4)
    nxt = opt.ask()
    assert all(0.0 <= v <= 1.0 for v in nxt)


def test_warped_device_chain_equals_oracle_chain(O):
    """warp_inputs=True through the sampler's asynchronous path (bgp_lml_batch_warped_submit / _wait: the per-walker
    Beta parameters are extra chain columns, bask/bayesgpr.py:353-365): the device-driven chain must coincide with the
    per-walker oracle replay (warp -> LML, default priors + N(0, 0.3) warp priors) on the same RandomState stream --
    one accept/reject flip would make them diverge."""
    import scipy.stats as st

    import bayes_skopt_amd as bask

    rng = np.random.RandomState(4)
    n, d, W, steps = 50, 2, 20, 25
    X = rng.uniform(size=(n, d))
    y = np.sin(9.0 * X[:, 0] ** 2) + X[:, 1] + 0.05 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    ad = np.full(n, 1e-10)
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=7, warp_inputs=True,
                       resident_sampler=False)  # (the host-driven loop: the resident form has its own test, test_gpu_resident.py)
    gp.fit(X, y, n_desired_samples=W, n_burnin=0, n_walkers_per_thread=W, progress=False)
    start = np.array(gp.pos_, copy=True)
    assert start.shape == (W, d + 2 + 2 * d)
    submitted = []
    real = gp._ctx.lml_warped_submit

    def spy(H, Wp):
        ok = real(H, Wp)
        submitted.append((len(H), ok))
        return ok

    gp._ctx.lml_warped_submit = spy
    gp.random_state = np.random.RandomState(321)
    gp.sample(n_desired_samples=W * steps, n_burnin=0, n_walkers_per_thread=W, position=start)
    dev_chain = gp._sampler.get_chain()
    assert len(submitted) == 1 + 2 * steps
    assert all(ok for b, ok in submitted if b == W // 2)  # every half-step block went through submit / wait
    seed = np.random.RandomState(321).randint(0, np.iinfo(np.int32).max)
    wp = st.norm(loc=0.0, scale=0.3).logpdf

    def log_prob(theta):
        th, w = theta[: d + 2], theta[d + 2:]
        lp = float(O.default_log_prior(th[None, :], d)[0]) + float(np.sum(wp(w)))
        lp += O.lml_warped(X, y, ad, th, w)
        return lp if np.isfinite(lp) else -np.inf

    ref_chain, ref_lp, _, _, _ = O.stretch_move_sampler(log_prob, start, steps, np.random.RandomState(seed))
    np.testing.assert_allclose(dev_chain, ref_chain, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(gp._sampler.get_log_prob(), ref_lp, rtol=1e-7)
