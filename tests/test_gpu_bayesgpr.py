"""End-to-end GPU tests of the bask API surface: the reference's own behavioural tests
(tests/test_bayesgpr.py, tests/test_utils.py of kiudee/bayes-skopt) re-expressed on this build, plus
LML parity at every theta the sampler visits."""
import numpy as np
import pytest
from scipy.stats import halfnorm, invgamma

from conftest import assert_variance_close, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bask():
    import bayes_skopt_amd as bask

    assert bask._lib.device_count() >= 1
    return bask


@pytest.fixture
def minimal_gp(bask):
    from bayes_skopt_amd.kernels import RBF, ConstantKernel

    kernel = ConstantKernel(constant_value=1**2, constant_value_bounds=(0.01**2, 1**2)) * RBF(
        length_scale=1.0, length_scale_bounds=(0.5, 1.5)
    )
    return bask.BayesGPR(random_state=1, normalize_y=False, kernel=kernel)


@pytest.fixture
def minimal_priors():
    return [
        lambda x: halfnorm(scale=1.0).logpdf(np.sqrt(np.exp(x))) + x / 2.0 - np.log(2.0),
        lambda x: invgamma(a=5.0, scale=1.0).logpdf(np.exp(x)) + x,
        lambda x: halfnorm(scale=1.0).logpdf(np.sqrt(np.exp(x))) + x / 2.0 - np.log(2.0),
    ]


def test_noise_vector(minimal_gp, minimal_priors):
    """reference tests/test_bayesgpr.py:36-51"""
    X = np.array([[0.0], [0.0]])
    y = np.array([1.0, 0.0])
    noise_vector = np.array([1234, 0.0])
    minimal_gp.fit(X, y, noise_vector=noise_vector, n_burnin=1, progress=False, priors=minimal_priors)
    prediction = minimal_gp.predict(np.array([[0.0]]))
    assert prediction < 0.01


def test_noise_set_to_zero(minimal_gp, minimal_priors):
    """reference tests/test_bayesgpr.py:54-62"""
    X = np.array([[0.1], [0.0], [-0.1]])
    y = np.array([0.0, 0.0, 0.0])
    minimal_gp.fit(X, y, n_burnin=1, progress=False, priors=minimal_priors)
    minimal_gp.theta = np.array([0.0, 0.0, 0.0])
    assert minimal_gp.predict(np.array([[0.0]]), return_std=True)[1] >= 1.0
    with minimal_gp.noise_set_to_zero():
        assert minimal_gp.predict(np.array([[0.0]]), return_std=True)[1] < 1.0
    assert minimal_gp.predict(np.array([[0.0]]), return_std=True)[1] >= 1.0


def test_sample_without_fit(minimal_gp):
    """reference tests/test_bayesgpr.py:65-68"""
    with pytest.raises(ValueError):
        minimal_gp.sample()


def test_fit_sample_predict_against_oracle(bask):
    """fit (MAP + MCMC) on config-A-sized data; every quantity derived from the final theta must
    match the oracle evaluated at that theta; the chain has the reference's shape."""
    from oracle import gp_oracle as O

    n, d = 128, 2
    X, y = synth(n, d, 0)
    kernel = bask.utils.construct_default_kernel(list(range(d)))
    gp = bask.BayesGPR(kernel=kernel, random_state=3, normalize_y=False)
    gp.fit(X, y, n_desired_samples=200, n_burnin=5, n_walkers_per_thread=40, progress=False)
    assert gp.chain_.shape == (200, d + 2)  # ceil(200/40) kept steps x 40 walkers
    assert gp.pos_.shape == (40, d + 2)
    th = gp.theta
    assert np.all(np.isfinite(th))
    ad = np.full(n, 1e-10)
    np.testing.assert_allclose(gp.log_marginal_likelihood_value_, O.lml(X, y, ad, th), rtol=1e-6)
    Lo, Kio, ao = O.posterior(X, y, ad, th)
    np.testing.assert_allclose(gp.alpha_, ao, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gp.L_, Lo, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(gp.K_inv_, Kio, rtol=1e-6, atol=1e-8 * np.abs(Kio).max())
    Xq = np.random.RandomState(5).uniform(size=(33, d))
    mean, std = gp.predict(Xq, return_std=True)
    mo, so = O.predict(X, y, ad, th, Xq)
    np.testing.assert_allclose(mean, mo, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(std, so, rtol=1e-6, atol=1e-8)
    with gp.noise_set_to_zero():
        m0, s0 = gp.predict(Xq, return_std=True)
    mo0, so0 = O.predict(X, y, ad, th, Xq, noise_zero=True)
    np.testing.assert_allclose(m0, mo0, rtol=1e-6, atol=1e-9)
    assert_variance_close(s0**2, so0**2, O.predict_variance_selfdiff(X, y, ad, th, Xq, noise_zero=True))
    # warm start: a second sample() resumes from pos_ and keeps the walker count
    gp.sample(n_desired_samples=80, n_burnin=0, n_walkers_per_thread=40)
    assert gp.chain_.shape == (80, d + 2)
    # the MAP start lies inside the kernel bounds and the median is a plausible posterior point
    assert 0.05 < np.exp(th[1]) < 5.0


def test_sampler_visits_have_oracle_lml(bask):
    """LML parity at every theta the device sampler visited (statistical chain parity is unpinned:
    emcee is absent -- see oracle/gp_oracle.py)."""
    from oracle import gp_oracle as O

    n, d = 96, 3
    X, y = synth(n, d, 7)
    kernel = bask.utils.construct_default_kernel(list(range(d)))
    gp = bask.BayesGPR(kernel=kernel, random_state=11)
    gp.fit(X, y, n_desired_samples=60, n_burnin=2, n_walkers_per_thread=20, progress=False)
    chain = gp._sampler.get_chain(flat=True)
    lps = gp._sampler.get_log_prob(flat=True)
    idx = np.random.RandomState(0).choice(len(chain), size=25, replace=False)
    ad = np.full(n, 1e-10)
    for i in idx:
        ref = O.lml(X, y, ad, chain[i]) + float(O.default_log_prior(chain[i][None, :], d)[0])
        np.testing.assert_allclose(lps[i], ref, rtol=1e-6)


def test_normalize_y_and_noise_vector_scaling(bask):
    """noise_vector is divided by std(y)^2 under normalize_y (bask/bayesgpr.py:480-483,603-605)."""
    n, d = 40, 2
    X, y = synth(n, d, 8)
    y = 5.0 * y + 3.0
    kernel = bask.utils.construct_default_kernel(list(range(d)))
    gp = bask.BayesGPR(kernel=kernel, random_state=2, normalize_y=True)
    nv = np.full(n, 0.5)
    gp.fit(X, y, noise_vector=nv, n_desired_samples=40, n_burnin=1, n_walkers_per_thread=20, progress=False)
    np.testing.assert_allclose(gp.alpha, 1e-10 + nv / np.std(y) ** 2)
    pred = gp.predict(X[:5])
    assert np.all(np.abs(pred - y[:5]) < 4 * np.std(y))


def test_device_chain_equals_oracle_chain(bask):
    """Same data, same priors, same RandomState stream: the chain produced by the batched device log-prob
    (BayesGPR.sample) must coincide with the chain of the per-walker CPU restatement (oracle sampler +
    oracle LML).  A single accept/reject flip would make the trajectories diverge grossly, so agreement
    to 1e-8 over 40 steps x 16 walkers certifies LML parity at every visited theta AND identical sampler
    semantics (start ball, RNG consumption, acceptance rule, chain layout)."""
    from oracle import gp_oracle as O

    n, d, W, steps = 60, 2, 16, 40
    X, y = synth(n, d, 17)
    ad = np.full(n, 1e-10)
    kernel = bask.utils.construct_default_kernel(list(range(d)))
    gp = bask.BayesGPR(kernel=kernel, random_state=5)
    gp.fit(X, y, n_desired_samples=W, n_burnin=0, n_walkers_per_thread=W, progress=False)  # MAP + 1 step
    start = np.array(gp.pos_, copy=True)
    # device-driven continuation, with a known sampler seed
    gp.random_state = np.random.RandomState(123)
    gp.sample(n_desired_samples=W * steps, n_burnin=0, n_walkers_per_thread=W, position=start)
    dev_chain = gp._sampler.get_chain()
    # oracle-driven replay: BayesGPR.sample seeds its sampler with RandomState(random_state.randint(0, 2^31-1))
    seed = np.random.RandomState(123).randint(0, np.iinfo(np.int32).max)

    def log_prob(theta):
        lp = float(O.default_log_prior(theta[None, :], d)[0]) + O.lml(X, y, ad, theta)
        return lp if np.isfinite(lp) else -np.inf

    ref_chain, ref_lp, _, _, _ = O.stretch_move_sampler(log_prob, start, steps, np.random.RandomState(seed))
    assert dev_chain.shape == ref_chain.shape == (steps, W, d + 2)
    np.testing.assert_allclose(dev_chain, ref_chain, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(gp._sampler.get_log_prob(), ref_lp, rtol=1e-8)
    np.testing.assert_allclose(gp.theta, O.geometric_median(ref_chain.reshape(-1, d + 2)), rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("kind", ["matern52", "matern32", "rbf", "sum_matern12"])
def test_predict_gradients_match_finite_differences(bask, kind):
    """predict(return_mean_grad, return_std_grad) (skopt's predict, forwarded at bask/bayesgpr.py:633): the
    analytic gradients at one query point against central differences of the device predict itself."""
    from bayes_skopt_amd.kernels import RBF, ConstantKernel, Matern

    rng = np.random.RandomState(3)
    X = rng.uniform(size=(60, 3))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.05 * rng.randn(60)
    kernel = {
        "matern52": ConstantKernel(1.0, (0.1, 2.0)) * Matern([0.4, 0.3, 0.5], (0.2, 0.8), nu=2.5),
        "matern32": ConstantKernel(1.0, (0.1, 2.0)) * Matern(0.4, (0.2, 0.8), nu=1.5),
        "rbf": ConstantKernel(1.0, (0.1, 2.0)) * RBF([0.4, 0.3, 0.5], (0.2, 0.8)),
        "sum_matern12": ConstantKernel(0.5, (0.1, 2.0)) + Matern(0.6, (0.2, 0.9), nu=0.5),
    }[kind]
    gp = bask.BayesGPR(kernel=kernel, random_state=0, normalize_y=True)
    gp.fit(X, y, n_desired_samples=40, n_burnin=5, n_walkers_per_thread=20, progress=False)
    x0 = np.array([[0.37, 0.52, 0.61]])
    with gp.noise_set_to_zero():
        mu, std, gmu, gstd = gp.predict(x0, return_std=True, return_mean_grad=True, return_std_grad=True)
        mu2, gmu2 = gp.predict(x0, return_mean_grad=True)
        h = 1e-5
        fd_mu, fd_std = np.zeros(3), np.zeros(3)
        for k in range(3):
            e = np.zeros(3)
            e[k] = h
            mp, sp = gp.predict(x0 + e, return_std=True)
            mm, sm = gp.predict(x0 - e, return_std=True)
            fd_mu[k] = (mp[0] - mm[0]) / (2 * h)
            fd_std[k] = (sp[0] - sm[0]) / (2 * h)
    assert gmu.shape == (3,) and gstd.shape == (3,)
    np.testing.assert_allclose(gmu, gmu2)
    np.testing.assert_allclose(gmu, fd_mu, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(gstd, fd_std, rtol=2e-4, atol=1e-6)
    with pytest.raises(NotImplementedError):
        gp.predict(np.vstack([x0, x0]), return_mean_grad=True)
    with pytest.raises(ValueError):
        gp.predict(x0, return_std_grad=True)


def test_sklearn_estimator_protocol_pickle_and_score(bask):
    """The reference's BayesGPR is a scikit-learn estimator (it subclasses skopt's / sklearn's
    GaussianProcessRegressor): get_params / set_params / clone / score work, and a fitted model survives pickle
    and deepcopy -- the ctypes device context does not travel and is rebuilt from the stored training data."""
    import copy
    import pickle

    from sklearn.base import clone

    from bayes_skopt_amd.kernels import ConstantKernel, Matern

    rng = np.random.RandomState(0)
    X = rng.uniform(size=(70, 2))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.05 * rng.randn(70)
    gp = bask.BayesGPR(kernel=ConstantKernel(1.0, (0.1, 2.0)) * Matern([0.4, 0.4], (0.2, 0.8), nu=2.5), normalize_y=True,
                       random_state=2)
    assert {"kernel", "alpha", "normalize_y", "warp_inputs", "noise", "random_state"} <= set(gp.get_params(deep=False))
    fresh = clone(gp)
    assert fresh.chain_ is None and fresh.normalize_y is True
    gp.fit(X, y, n_desired_samples=40, n_burnin=5, n_walkers_per_thread=20, progress=False)
    Xq = rng.uniform(size=(25, 2))
    mu, std = gp.predict(Xq, return_std=True)
    r2 = gp.score(X, y)
    assert 0.9 < r2 <= 1.0
    for other in (pickle.loads(pickle.dumps(gp)), copy.deepcopy(gp)):
        assert other._ctx_obj is None          # nothing device-side travelled
        np.testing.assert_array_equal(other.chain_, gp.chain_)
        mu2, std2 = other.predict(Xq, return_std=True)   # rebuilds the context and the posterior on demand
        np.testing.assert_array_equal(mu2, mu)
        np.testing.assert_array_equal(std2, std)
        np.testing.assert_allclose(other.K_inv_, gp.K_inv_, rtol=0, atol=0)
        assert other.score(X, y) == r2
        other.sample(n_desired_samples=40, n_burnin=2, n_walkers_per_thread=20)  # MCMC resumes from pos_
        assert other.chain_.shape == gp.chain_.shape
