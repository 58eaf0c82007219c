"""The reference's tests/test_optimizer.py and tests/test_acquisition.py re-expressed on this build
(GPU: every tell() runs the device MCMC + acquisition kernels)."""
import numpy as np
import pytest
from numpy.testing import assert_almost_equal, assert_equal
from scipy.stats import halfnorm, invgamma

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bask():
    import bayes_skopt_amd as bask

    assert bask._lib.device_count() >= 1
    return bask


def bench1(x):
    """skopt.benchmarks.bench1: x[0]**2"""
    return x[0] ** 2


def test_multiple_asks(bask):
    """reference tests/test_optimizer.py:14-26"""
    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=1)
    opt.run(bench1, n_iter=3, gp_burnin=0, n_samples=1)
    assert_equal(len(opt.Xi), 3)
    opt.ask()
    assert_equal(len(opt.Xi), 3)
    assert_equal(opt.ask(), opt.ask())


@pytest.mark.parametrize("init_strategy", ("r2", "sb", "random"))
def test_initial_points(bask, init_strategy):
    """reference tests/test_optimizer.py:28-47"""
    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=3, init_strategy=init_strategy)
    x = opt.ask()
    assert not isinstance(x[0], list)
    opt.tell([x], [0.0])
    assert opt._n_initial_points == opt.n_initial_points_ - 1
    opt.tell([x], [0.0])
    assert opt._n_initial_points == opt.n_initial_points_ - 2
    assert opt.gp.chain_ is None
    opt.tell([[0.1], [0.2], [0.3]], [0.0, 0.1, 0.2], replace=True)
    assert opt._n_initial_points == opt.n_initial_points_ - 3
    assert opt.gp.chain_ is not None


def test_noise_vector(bask):
    """reference tests/test_optimizer.py:49-64"""
    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=5)
    opt.tell(
        [[-2.0], [-1.0], [0.0], [1.0], [2.0]],
        [0.0, -1.0, 0.0, -1.0, 0.0],
        noise_vector=[1.0, 1.0, 1.0, 0.0, 1.0],
    )
    y_noisy, y = opt.gp.predict(opt.space.transform([[-1.0], [1.0]]))
    assert y_noisy > y
    x = opt.ask()
    opt.tell(x, 0.0, noise_vector=0.5)


def test_run_with_noise(bask):
    """reference tests/test_optimizer.py:66-73"""
    random_state = np.random.RandomState(123)

    def func(x):
        return (np.sin(x) + random_state.randn()).item(), 1.0

    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=1)
    opt.run(func, n_iter=2, n_samples=1, gp_burnin=0)
    assert_almost_equal(opt.gp.alpha, np.ones(2))


def test_no_error_on_unknown_kwargs(bask):
    bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=5, unknown_argument=42)


def test_error_on_invalid_priors(bask):
    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], gp_priors=[], n_initial_points=0)
    with pytest.raises(ValueError):
        opt.tell([(0.0,)], 0.0)


@pytest.fixture
def fitted_minimal_gp(bask):
    from bayes_skopt_amd.kernels import RBF, ConstantKernel

    kernel = ConstantKernel(constant_value=1**2, constant_value_bounds=(0.01**2, 1**2)) * RBF(
        length_scale=1.0, length_scale_bounds=(0.5, 1.5)
    )
    gp = bask.BayesGPR(random_state=1, normalize_y=False, kernel=kernel)
    priors = [
        lambda x: halfnorm(scale=1.0).logpdf(np.sqrt(np.exp(x))) + x / 2.0 - np.log(2.0),
        lambda x: invgamma(a=5.0, scale=1.0).logpdf(np.exp(x)) + x,
        lambda x: halfnorm(scale=1.0).logpdf(np.sqrt(np.exp(x))) + x / 2.0 - np.log(2.0),
    ]
    x = np.array([-2.0, -1.0, 1.0, 2.0])[:, None]
    y = np.array([0, -1, 1, 2])
    gp.fit(x, y, priors=priors, progress=False, n_burnin=1)
    return gp


# (acquisition, n_samples, argmax pinned by the reference's tests/test_acquisition.py:42-53).
# These indices run through the MAP fit, emcee's exact RNG stream (fit with random_state=1), the
# geometric median, the hyper-sample selection and -- ThompsonSampling, and the Thompson points of PVRS --
# numpy's legacy SVD multivariate normal behind sample_y: reproducing ALL EIGHT exactly pins this build's
# restatement of the ensemble sampler (emcee itself is absent from the image) and its reference-variate
# function draws (BayesGPR(mvn="auto"): device mean / covariance, host SVD draw up to 512 query points)
# against the reference's own test vectors.
ACQ_CASES = [
    ("VarianceReduction", 0, 50),
    ("PVRS", 0, 38),
    ("LCB", 1, 38),
    ("ExpectedImprovement", 1, 33),
    ("Expectation", 1, 30),
    ("TopTwoEI", 1, 32),
    ("MaxValueSearch", 1, 37),
    ("ThompsonSampling", 1, 25),
]


@pytest.mark.parametrize("name, n_samples, expected", ACQ_CASES)
def test_acquisition_argmax_matches_reference_pins(bask, fitted_minimal_gp, name, n_samples, expected):
    x = np.linspace(-2.0, 2.0, num=101)[:, None]
    acq = bask.acquisition.evaluate_acquisitions(
        X=x, gpr=fitted_minimal_gp, acquisition_functions=[getattr(bask.acquisition, name)()], random_state=1,
        n_samples=n_samples,
    )
    got = int(np.argmax(acq))
    print(f"{name}: argmax {got} (reference pins {expected})")
    assert got == expected


@pytest.mark.parametrize("name, n_samples", [("ThompsonSampling", 1)])
def test_acquisition_sampling_based_runs(bask, fitted_minimal_gp, name, n_samples, monkeypatch):
    """The device draw (Cholesky factor, other variates than the reference's SVD draw): same distribution, so the same
    qualitative answer."""
    monkeypatch.setattr(fitted_minimal_gp, "mvn", "cholesky")
    x = np.linspace(-2.0, 2.0, num=101)[:, None]
    acq = bask.acquisition.evaluate_acquisitions(
        X=x, gpr=fitted_minimal_gp, acquisition_functions=[getattr(bask.acquisition, name)()], random_state=1,
        n_samples=n_samples,
    )
    assert acq.shape == (1, 101) and np.all(np.isfinite(acq))
    print(f"{name}: argmax {int(np.argmax(acq))}")
    # the data decrease towards x = -1 (y = -1): every criterion must prefer the left half
    assert int(np.argmax(acq)) < 60


def test_evaluate_acquisitions_batched_equals_per_sample_loop(bask, fitted_minimal_gp):
    """The batched posterior build + predict must equal the reference's per-draw loop
    (gpr.theta = chain_[i]; predict inside noise_set_to_zero; bask/acquisition.py:112-141)."""
    gp = fitted_minimal_gp
    x = np.linspace(-2.0, 2.0, num=41)[:, None]
    acqs = [bask.acquisition.ExpectedImprovement(), bask.acquisition.LCB()]
    out = bask.acquisition.evaluate_acquisitions(X=x, gpr=gp, acquisition_functions=acqs, random_state=3, n_samples=5)
    rs = np.random.RandomState(3)
    idx = rs.choice(len(gp.chain_), replace=False, size=5)
    theta_backup = gp.theta
    ref = np.zeros((2, 41))
    for i in idx:
        gp.theta = gp.chain_[i]
        with gp.noise_set_to_zero():
            mu, std = gp.predict(x, return_std=True)
        for j, a in enumerate(acqs):
            ref[j] += a(mu, std) / 5
    gp.theta = theta_backup
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(gp.theta, theta_backup)


def test_tell_loop_pvrs_small(bask):
    """Config-E-shaped loop at small size: several tells with PVRS over a candidate grid."""
    rng = np.random.RandomState(0)
    opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * 3, n_points=300, n_initial_points=8, init_strategy="r2",
                         acq_func="pvrs", random_state=0)

    def f(x):
        return float(np.sin(3 * np.sum(x)) + 0.05 * rng.randn())

    for _ in range(11):
        x = opt.ask()
        res = opt.tell(x, f(x), gp_samples=100, gp_burnin=2, n_samples=0)
    assert len(opt.Xi) == 11 and opt.gp.chain_.shape == (100, 5)
    assert opt.gp.pos_.shape == (100, 5)  # 100 walkers through the Optimizer (bask/bayesgpr.py:390)
    nxt = opt.ask()
    assert len(nxt) == 3 and all(0.0 <= v <= 1.0 for v in nxt)
    assert res.fun == min(opt.yi)


# ---- post-hoc diagnostics (tests/test_optimizer.py:85-175 of the reference).  The reference pins its numbers to
# two decimals through emcee's stream and numpy's SVD-based MVN draws (ONE realisation of 200 / 100 function draws).
# (1) With the reference's own fixture -- ONE RandomState(123) shared by the Optimizer and the diagnostic call -- and
#     reference-variate draws (BayesGPR.mvn = "reference": device mean / covariance, numpy's legacy SVD draw on the host)
#     the pins are reproduced as the reference asserts them (assert_almost_equal, decimal=2): 0.99 / (0.98, 0.86) / 1.00 and
#     0.297 / 0.247 / 0.279 on the MI355X.
# (2) The device draws (Cholesky factor: same distribution, other variates) are checked statistically: the Monte-Carlo
#     error on this side is driven down -- 2000 draws for the probabilities, the mean over twelve seeds of the reference's
#     own coarse estimator settings for the gap -- and the results must sit within 0.03 of the pins (what is left is the error of the reference's single realisation).
def _reference_fixture_optimizer(bask):
    rs = np.random.RandomState(123)  # tests/test_optimizer.py:9-11 of the reference
    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=0, random_state=rs)
    opt.tell([[-2.0], [-1.0], [0.0], [1.0], [2.0]], [2.0, 0.0, -2.0, 0.0, 2.0], gp_burnin=10)
    opt.gp.mvn = "reference"  # (501 query points: beyond what "auto" hands to the host SVD)
    return opt, rs


@pytest.mark.parametrize(
    "kw,expected",
    [
        (dict(normalized_scores=False, threshold=1.0), 0.99),
        (dict(normalized_scores=False, threshold=(0.9, 0.5)), (0.98, 0.86)),
        (dict(normalized_scores=True, threshold=1.0), 0.99),
    ],
)
def test_probability_of_optimality_reproduces_the_reference_pins(bask, kw, expected):
    """tests/test_optimizer.py:85-110 of the reference, as written there."""
    opt, rs = _reference_fixture_optimizer(bask)
    prob = opt.probability_of_optimality(threshold=kw["threshold"], n_random_starts=100, random_state=rs,
                                         normalized_scores=kw["normalized_scores"])
    np.testing.assert_almost_equal(prob, expected, decimal=2)


@pytest.mark.parametrize(
    "kw,expected",
    [
        (dict(normalized_scores=False, use_mean_gp=True), 0.3),
        (dict(normalized_scores=True, use_mean_gp=True), 0.25),
        (dict(normalized_scores=True, use_mean_gp=False), 0.29),
    ],
)
def test_expected_optimality_gap_reproduces_the_reference_pins(bask, kw, expected):
    """tests/test_optimizer.py:113-141 of the reference, as written there."""
    opt, rs = _reference_fixture_optimizer(bask)
    gap = opt.expected_optimality_gap(random_state=rs, n_probabilities=10, n_space_samples=100, n_gp_samples=100,
                                      n_random_starts=10, tol=0.1, use_mean_gp=kw["use_mean_gp"],
                                      normalized_scores=kw["normalized_scores"])
    np.testing.assert_almost_equal(gap, expected, decimal=2)


def _five_point_optimizer(bask, seed):
    opt = bask.Optimizer(dimensions=[(-2.0, 2.0)], n_initial_points=0, random_state=np.random.RandomState(seed))
    opt.tell([[-2.0], [-1.0], [0.0], [1.0], [2.0]], [2.0, 0.0, -2.0, 0.0, 2.0], gp_burnin=10)
    opt.gp.mvn = "cholesky"  # the device draws: checked statistically below
    return opt


@pytest.mark.parametrize(
    "kw,expected",
    [
        (dict(normalized_scores=False, threshold=1.0), 0.99),
        (dict(normalized_scores=False, threshold=(0.9, 0.5)), (0.98, 0.86)),
        (dict(normalized_scores=True, threshold=1.0), 0.99),
    ],
)
def test_probability_of_optimality(bask, kw, expected):
    opt = _five_point_optimizer(bask, 0)
    prob = opt.probability_of_optimality(threshold=kw["threshold"], n_random_starts=100, n_gp_samples=2000,
                                         random_state=np.random.RandomState(0),
                                         normalized_scores=kw["normalized_scores"])
    np.testing.assert_allclose(prob, expected, atol=0.03)
    assert np.all(np.asarray(prob) <= 1.0) and np.all(np.asarray(prob) >= 0.0)


def test_probability_of_optimality_is_monotone_in_the_threshold(bask):
    opt = _five_point_optimizer(bask, 1)
    p = opt.probability_of_optimality(threshold=[0.0, 0.25, 0.5, 1.0, 2.0], n_random_starts=20, random_state=3,
                                      normalized_scores=False)
    assert all(b >= a for a, b in zip(p, p[1:]))
    assert p[-1] > 0.97


@pytest.mark.parametrize(
    "kw,expected",
    [
        (dict(normalized_scores=False, use_mean_gp=True), 0.3),
        (dict(normalized_scores=True, use_mean_gp=True), 0.25),
        (dict(normalized_scores=True, use_mean_gp=False), 0.29),
    ],
)
def test_expected_optimality_gap(bask, kw, expected):
    opt = _five_point_optimizer(bask, 0)
    gaps = [opt.expected_optimality_gap(random_state=np.random.RandomState(seed), n_probabilities=10, n_space_samples=100,
                                        n_gp_samples=100, n_random_starts=10, tol=0.1, use_mean_gp=kw["use_mean_gp"],
                                        normalized_scores=kw["normalized_scores"]) for seed in range(12)]
    assert all(0.0 < g < 1.0 for g in gaps)
    np.testing.assert_allclose(np.mean(gaps), expected, atol=0.03)


def test_optimum_intervals(bask):
    opt = bask.Optimizer(dimensions=[(0.0, 1.0)], random_state=0, acq_func="mean", n_points=100)
    x = np.linspace(0, 1, num=20)[:, None]
    y = np.cos(np.pi * 4 * x).flatten() + opt.rng.randn(20) * 0.1
    opt.tell(x.tolist(), y.tolist(), gp_burnin=20, progress=False, n_samples=1)
    intervals = opt.optimum_intervals(random_state=0, space_samples=100)
    assert len(intervals) == 1
    assert len(intervals[0]) >= 2          # cos(4 pi x) has two minima on [0, 1]
    assert len(intervals[0][0]) == 2
    centres = sorted(np.mean(iv) for iv in intervals[0])
    assert abs(centres[0] - 0.25) < 0.1 and abs(centres[-1] - 0.75) < 0.1
    intervals = opt.optimum_intervals(random_state=0, space_samples=100, multimodal=False)
    assert len(intervals) == 1
    assert len(intervals[0]) == 2
    opt_cat = bask.Optimizer(dimensions=[(0.0, 1.0), ["a", "b"]], n_initial_points=2)
    with pytest.raises(NotImplementedError):
        opt_cat.optimum_intervals()


def test_optimizer_survives_pickle(bask):
    """A tuning run checkpointed with pickle continues where it stopped (the GP's device context is rebuilt)."""
    import pickle

    rng = np.random.RandomState(0)
    opt = bask.Optimizer(dimensions=[(0.0, 1.0), (0.0, 1.0)], n_initial_points=4, random_state=0, n_points=200)
    for _ in range(6):
        x = opt.ask()
        opt.tell(x, float(np.sin(5 * x[0]) + x[1] ** 2 + 0.01 * rng.randn()), gp_samples=40, gp_burnin=3)
    clone = pickle.loads(pickle.dumps(opt))
    assert clone.Xi == opt.Xi and clone.yi == opt.yi
    np.testing.assert_array_equal(clone.gp.chain_, opt.gp.chain_)
    x = clone.ask()
    res = clone.tell(x, 0.3, gp_samples=40, gp_burnin=3)
    assert len(res.x_iters) == 7 and clone.gp.X_train_.shape[0] == 7
    Xq = rng.uniform(size=(5, 2))
    assert np.all(np.isfinite(clone.gp.predict(Xq)))
