"""Device side of evaluate_acquisitions (bask/acquisition.py:112-139): the closed-form criteria
(ExpectedImprovement :154-172, Expectation :197-201, LCB :204-216) evaluated and averaged over the hyper-posterior
draws on the GPU, against the reference classes' golden outputs and against the host loop over draws."""
import numpy as np
import pytest

from conftest import load_golden, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import bayes_skopt_amd  # noqa: F401
    from bayes_skopt_amd import _lib

    assert _lib.device_count() >= 1
    return _lib


@pytest.fixture(scope="module")
def ctx(lib):
    X, y = synth(64, 2, 5)
    c = lib.Context(X, y, 1e-10, max_batch=4)
    yield c
    c.close()


def test_device_closed_forms_match_the_reference_classes(lib, ctx):
    """Same inputs and golden outputs as tests/test_cpu_host.py::test_closed_form_acquisitions_match_the_reference_classes
    (tests/golden/reference_tier1.npz: outputs of the reference's bask.acquisition classes)."""
    g = load_golden("reference_tier1.npz")
    mu, std = g["acq_mu"], g["acq_std"]
    kinds = [lib.ACQ_EI, lib.ACQ_EI, lib.ACQ_LCB, lib.ACQ_LCB, lib.ACQ_MEAN, lib.ACQ_STD]
    params = [np.nan, -0.3, 1.96, 3.0, 0.0, 0.0]
    out = ctx.acq_values(mu[None, :], std[None, :], kinds, params, 1)
    # x Phi(x) + phi(x) cancels to ~1/x^2 of its terms in the far lower tail (values down to 1e-299 here): one ulp of
    # erfc between libm and the device library shows up as ~x^2 ulp, hence 1e-9 and not 1e-13 for EI
    np.testing.assert_allclose(out[0], g["acq_ei"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(out[1], g["acq_ei_yopt"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(out[2], g["acq_lcb"], rtol=1e-13)
    np.testing.assert_allclose(out[3], g["acq_lcb3"], rtol=1e-13)
    np.testing.assert_allclose(out[4], g["acq_mean"], rtol=1e-13)
    np.testing.assert_array_equal(out[5], std)


def test_device_average_over_draws_follows_the_reference_rules(lib, ctx):
    """Draw order, / n_samples, zero where std <= 0, and "a draw with a non-finite value contributes nothing"
    (bask/acquisition.py:137-139), for EI in both tails of Phi."""
    from bayes_skopt_amd import acquisition as acq

    rng = np.random.RandomState(0)
    B, m, n_samples = 7, 333, 9  # (n_samples != B: the divisor is the requested count)
    mu = rng.randn(B, m) * np.array([1, 1, 5, 1, 1, 30, 1])[:, None]
    std = np.abs(rng.randn(B, m)) * np.array([1, 0.01, 1, 1, 1, 0.05, 1])[:, None] + 1e-3
    std[0, ::7] = 0.0        # EI: exactly zero there
    mu[3, 11] = np.nan       # draw 3: EI (y_opt = NaN), LCB and mean all dropped; std kept
    std[4, 5] = np.inf       # draw 4: LCB / std dropped; EI: x = 0 -> 0.3989 * inf = inf -> dropped too
    objs = [acq.ExpectedImprovement(), acq.LCB(), acq.Expectation(), acq.LCB(), acq.ExpectedImprovement()]
    kws = [{}, {"alpha": 2.5}, {}, {"alpha": "inf"}, {"y_opt": 0.25}]
    kinds = [lib.ACQ_EI, lib.ACQ_LCB, lib.ACQ_MEAN, lib.ACQ_STD, lib.ACQ_EI]
    params = [np.nan, 2.5, 0.0, 0.0, 0.25]
    ref = np.zeros((len(objs), m))
    with np.errstate(all="ignore"):
        for b in range(B):
            for j, (o, kw) in enumerate(zip(objs, kws)):
                tmp = o(mu[b], std[b], **kw)
                if np.all(np.isfinite(tmp)):
                    ref[j] += tmp / n_samples
    out = ctx.acq_values(mu, std, kinds, params, n_samples)
    np.testing.assert_allclose(out[1:4], ref[1:4], rtol=1e-13)
    np.testing.assert_allclose(out[[0, 4]], ref[[0, 4]], rtol=1e-9, atol=1e-300)
    assert np.all(out[0, ::7] >= 0.0) and np.all(np.isfinite(out))


def test_device_acquisition_pass_equals_host_loop(lib):
    """evaluate_acquisitions with the device pass (build + predict + closed forms + average in one call, nothing but
    (n_acq, m) values returned) against the same call with the host loop over draws, at a size where the batched
    predict runs in two chunks of draws; a criterion without a device form (TopTwoEI) sends the whole call to the
    host path."""
    import bayes_skopt_amd as bask
    from bayes_skopt_amd import acquisition as acq

    n, d, m = 300, 4, 1500
    X, y = synth(n, d, 31)
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0, normalize_y=True)
    gp.fit(X, 3.0 * y + 1.5, n_desired_samples=60, n_burnin=5, progress=False)
    Xq = np.random.RandomState(32).uniform(size=(m, d))
    acqs = [acq.ExpectedImprovement(), acq.LCB(), acq.Expectation()]
    outs = {}
    for dev in (True, False):
        acq.DEVICE_ACQUISITIONS = dev
        try:
            outs[dev] = acq.evaluate_acquisitions(Xq, gp, acqs, n_samples=24, random_state=7, alpha=1.0)
        finally:
            acq.DEVICE_ACQUISITIONS = True
    np.testing.assert_allclose(outs[True], outs[False], rtol=1e-10, atol=1e-14)
    assert np.argmax(outs[True][0]) == np.argmax(outs[False][0])
    mixed = acq.evaluate_acquisitions(Xq, gp, acqs + [acq.TopTwoEI()], n_samples=24, random_state=7, alpha=1.0)
    np.testing.assert_array_equal(mixed[:3], outs[False])


def test_acq_entry_points_reject_bad_arguments(lib, ctx):
    mu = np.zeros((1, 4))
    with pytest.raises(Exception, match="kind"):
        ctx.acq_values(mu, mu + 1.0, [9], [0.0], 1)
    with pytest.raises(Exception, match="n_acq"):
        ctx.acq_values(mu, mu + 1.0, [lib.ACQ_EI] * (lib.ACQ_MAX + 1), [0.0] * (lib.ACQ_MAX + 1), 1)
