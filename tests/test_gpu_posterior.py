"""GPU parity of posterior build / predict / LML gradient / PVRS / sample_y through the C-ABI against
the golden vectors (sklearn 1.7.2; reference bask.acquisition via tier-1 import) and the oracle."""
import numpy as np
import pytest

from conftest import assert_variance_close, load_golden, synth

pytestmark = pytest.mark.gpu
RTOL = 1e-6


@pytest.fixture(scope="module")
def lib():
    import bayes_skopt_amd  # noqa: F401
    from bayes_skopt_amd import _lib

    assert _lib.device_count() >= 1
    return _lib


def test_posterior_factors_and_predict(lib):
    g = load_golden("predict.npz")
    for c in range(int(g["n_cases"])):
        pre = f"c{c}_"
        st, form = [str(s) for s in g[pre + "meta"]]
        X, y, Xq, ad, th = g[pre + "X"], g[pre + "y"], g[pre + "Xq"], g[pre + "alpha_diag"], g[pre + "theta"]
        ctx = lib.Context(X, y, ad, form=form, stationary=st, max_batch=4)
        res = ctx.posterior(th, want_L=True, want_alpha=True, want_K_inv=True)
        assert res["status"][0] == 0
        np.testing.assert_allclose(res["L"][0], g[pre + "L"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res["alpha"][0], g[pre + "alpha_vec"], rtol=1e-6, atol=1e-7)
        scale = np.abs(g[pre + "K_inv"]).max()
        np.testing.assert_allclose(res["K_inv"][0], g[pre + "K_inv"], rtol=1e-6, atol=1e-8 * scale)
        # predict with noise (kernel_ as built) and inside noise_set_to_zero (log s2 = -inf)
        mean, var, cov = ctx.predict(th, Xq, return_cov=True)
        np.testing.assert_allclose(mean[0], g[pre + "mean"], rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(np.sqrt(var[0]), g[pre + "std"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(cov[0], g[pre + "cov"], rtol=RTOL, atol=1e-8)
        th0 = th.copy()
        th0[-1] = -np.inf
        mean0, var0, cov0 = ctx.predict(th0, Xq, return_cov=True)
        np.testing.assert_allclose(mean0[0], g[pre + "mean0"], rtol=RTOL, atol=1e-9)
        # noise-free variance: 1e-6 relative, or 4 x the difference between the reference's own two formulas where that is larger
        assert_variance_close(var0[0], g[pre + "std0"] ** 2, g[pre + "var0_selfdiff"])
        assert_variance_close(var[0], g[pre + "std"] ** 2, g[pre + "var_selfdiff"])
        np.testing.assert_allclose(cov0[0], g[pre + "cov0"], rtol=RTOL, atol=1e-8)
        ctx.close()


def test_posterior_batch_of_hyper_samples(lib):
    """evaluate_acquisitions rebuilds the posterior once per hyper-posterior sample
    (bask/acquisition.py:112-121): here as ONE batched build + one batched predict."""
    from oracle import gp_oracle as O

    n, d, m, B = 150, 3, 70, 6
    X, y = synth(n, d, 21)
    Xq = np.random.RandomState(22).uniform(size=(m, d))
    base = np.array([0.0, -1.2, -1.0, -1.4, -4.0])
    TH = base + 0.2 * np.random.RandomState(23).randn(B, d + 2)
    ctx = lib.Context(X, y, 1e-10, max_batch=8)
    res = ctx.posterior(TH)
    assert np.all(res["status"] == 0)
    TH0 = TH.copy()
    TH0[:, -1] = -np.inf
    mean, var = ctx.predict(TH0, Xq)
    for b in range(B):
        mo, so = O.predict(X, y, np.full(n, 1e-10), TH[b], Xq, noise_zero=True)
        np.testing.assert_allclose(mean[b], mo, rtol=RTOL, atol=1e-9)
        assert_variance_close(var[b], so**2, O.predict_variance_selfdiff(X, y, np.full(n, 1e-10), TH[b], Xq, noise_zero=True))
        np.testing.assert_allclose(res["lml"][b], O.lml(X, y, np.full(n, 1e-10), TH[b]), rtol=RTOL)
    ctx.close()


def test_lml_gradient(lib):
    g = load_golden("lml_grad.npz")
    for c in range(int(g["n_cases"])):
        pre = f"c{c}_"
        st, form = [str(s) for s in g[pre + "meta"]]
        ctx = lib.Context(g[pre + "X"], g[pre + "y"], 1e-10, form=form, stationary=st, max_batch=4)
        val, grad, status = ctx.lml_grad(g[pre + "theta"])
        assert np.all(status == 0)
        np.testing.assert_allclose(val, g[pre + "lml"], rtol=RTOL)
        np.testing.assert_allclose(grad, g[pre + "grad"], rtol=1e-5, atol=1e-6)
        ctx.close()


def test_lml_gradient_many_dims_and_tiles(lib):
    """d > 16 (two staging passes) and n > 128 (several tiles, off-diagonal weight 2)."""
    from oracle import gp_oracle as O

    n, d = 300, 20
    X, y = synth(n, d, 31)
    th = np.concatenate([[0.1], np.log(0.5) + 0.1 * np.random.RandomState(1).randn(d), [np.log(0.05)]])
    ctx = lib.Context(X, y, 1e-10, max_batch=2)
    val, grad, _ = ctx.lml_grad(th)
    vo, go = O.lml_and_grad(X, y, np.full(n, 1e-10), th)
    np.testing.assert_allclose(val[0], vo, rtol=RTOL)
    np.testing.assert_allclose(grad[0], go, rtol=1e-5, atol=1e-6)
    ctx.close()


def test_pvrs_matches_reference_loop(lib):
    """bask.acquisition.PVRS per-candidate bordered Cholesky loop (golden: reference tier-1 import)
    vs the bordered-inverse identity on the device."""
    g = load_golden("reference_tier1.npz")
    for c in range(int(g["pvrs_cases"])):
        pre = f"pvrs{c}_"
        X, Xc, th = g[pre + "X"], g[pre + "Xc"], g[pre + "theta"]
        av = g[pre + "alpha_vec"]
        has_vec = av.size > 0
        y = np.zeros(len(X))
        ctx = lib.Context(X, y, av if has_vec else 1e-10, max_batch=4)
        assert ctx.pvrs_prepare(th, has_vec) == 0
        tp = Xc[np.argmin(g[pre + "thompson"], axis=0)]
        covs = ctx.pvrs(th, Xc, tp)
        np.testing.assert_allclose(covs, g[pre + "covs"], rtol=RTOL)
        if (pre + "vr") in g.files:  # VarianceReduction == PVRS with thompson points = all candidates
            vr = ctx.pvrs(th, Xc, Xc)
            np.testing.assert_allclose(vr, g[pre + "vr"], rtol=RTOL)
        ctx.close()


def test_sample_y_moments(lib):
    """Draws through the device Cholesky of the predictive covariance: f = mean + L z reproduces
    mean and cov exactly for the supplied z (checked against the oracle's mean / cov)."""
    from oracle import gp_oracle as O

    n, d, m = 80, 2, 150
    X, y = synth(n, d, 41)
    Xq = np.random.RandomState(42).uniform(size=(m, d))
    th = np.array([0.0, -1.0, -1.1, -3.0])
    ctx = lib.Context(X, y, 1e-10, max_batch=2)
    ctx.posterior(th)
    z = np.random.RandomState(43).standard_normal((5, m))
    jit = 1e-8
    out = ctx.sample_y(0, th, Xq, z, jitter=jit)
    mean, std, cov = O.predict(X, y, np.full(n, 1e-10), th, Xq, return_cov=True)
    Lc = np.linalg.cholesky(cov + jit * np.eye(m))
    np.testing.assert_allclose(out, mean[None, :] + z @ Lc.T, rtol=1e-6, atol=1e-7)
    ctx.close()


def test_sample_y_is_the_reference_distribution(lib):
    """a8 against the REFERENCE's own call: scikit-learn's GaussianProcessRegressor.sample_y (sklearn/_gpr.py:522-526, numpy's
    SVD-based multivariate normal; reached from bask/bayesgpr.py:669-678 with the noise switched off).  The device draws
    use a Cholesky factor and therefore other variates; what must agree is the distribution:
    (i) its parameters -- the device mean / covariance reconstructed from unit vectors -- with scikit-learn's
    predict(return_cov=True) at 1e-6, and (ii) sample mean and covariance of 4000 device draws with them within
    Monte-Carlo error, exactly as the 4000 reference draws stored in the golden file do."""
    g = load_golden("sample_y.npz")
    X, y, Xq, th = g["X"], g["y"], g["Xq"], g["theta"]
    m, nd = Xq.shape[0], int(g["ndraw"])
    ctx = lib.Context(X, y, 1e-10, max_batch=2)
    ctx.posterior(th)
    hk = th.copy()
    hk[-1] = -np.inf  # noise_set_to_zero: the white-noise level leaves the kernel, the factors stay
    # (i) parameters: z = 0 gives the mean, z = e_j the j-th column of the factor
    out = ctx.sample_y(0, hk, Xq, np.vstack([np.zeros((1, m)), np.eye(m)]), jitter=0.0)
    mean = out[0]
    Lf = (out[1:] - mean).T
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(Lf @ Lf.T, g["cov"], rtol=1e-6, atol=1e-9 * np.abs(g["cov"]).max())
    # (ii) 4000 draws: same Monte-Carlo agreement as the reference's own draws
    z = np.random.RandomState(11).standard_normal((nd, m))
    draws = ctx.sample_y(0, hk, Xq, z, jitter=0.0)
    sd = np.sqrt(np.diag(g["cov"]))
    zdev = np.abs(draws.mean(axis=0) - g["mean"]) / (sd / np.sqrt(nd))
    zref = np.abs(g["ref_sample_mean"] - g["mean"]) / (sd / np.sqrt(nd))
    assert zdev.max() < 4.5 and zref.max() < 4.5
    cdev, cref = np.cov(draws.T), g["ref_sample_cov"]
    scale = np.sqrt(np.outer(np.diag(g["cov"]), np.diag(g["cov"])))
    tol = 6.0 * np.sqrt(2.0 / nd)  # (a sample covariance entry scatters by ~ sqrt((1 + rho^2) / N) of the scale)
    assert np.abs(cdev - g["cov"]).max() / scale.max() < tol and np.abs(cref - g["cov"]).max() / scale.max() < tol
    ctx.close()


def _mvn_gp(g):
    """A BayesGPR whose median GP is the golden's: kernel 1.3 * Matern52([0.35, 0.5]) + White(0.02), y as given."""
    import bayes_skopt_amd as bask
    from bayes_skopt_amd.kernels import ConstantKernel, Matern

    X, y = g["X"], g["y"]
    gp = bask.BayesGPR(kernel=ConstantKernel(1.3) * Matern(length_scale=[0.35, 0.5], nu=2.5), normalize_y=False,
                       random_state=0)
    gp.fit(X, y, n_desired_samples=40, n_burnin=1, n_walkers_per_thread=20, progress=False)
    gp.theta = g["theta"]
    return gp


def test_sample_y_reference_variates_equal_sklearns_seeded_draws():
    """Row a8 with the reference's OWN variates (SURVEY 8c golden (8), tests/golden/mvn.npz): BayesGPR.sample_y(mvn="reference")
    -- mean and covariance from the device (bgp_predict_batch with cov), numpy's legacy SVD multivariate normal on the host,
    as sklearn/_gpr.py:522-526 behind bask/bayesgpr.py:669-678 -- returns scikit-learn's seeded draws: 1e-9 absolute at 6,
    24 and 64 query points, noise off (the reference's default) and on.  "auto" takes this mode up to 512 query points."""
    g = load_golden("mvn.npz")
    gp = _mvn_gp(g)
    for m in (6, 24, 64):
        Xq = g["Xq_%d" % m]
        for tag, noise in (("nf", False), ("ny", True)):
            want = g["%s_draws_%d" % (tag, m)]
            got = gp.sample_y(Xq, sample_mean=True, noise=noise, n_samples=int(g["ndraw"]), random_state=int(g["seed"]),
                              mvn="reference")
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
            auto = gp.sample_y(Xq, sample_mean=True, noise=noise, n_samples=int(g["ndraw"]), random_state=int(g["seed"]))
            np.testing.assert_array_equal(auto, got)
            # the device parameters themselves against scikit-learn's predict(return_cov=True)
            cm = gp.noise_set_to_zero() if not noise else None
            if cm is not None:
                with cm:
                    mean, cov = gp.predict(Xq, return_cov=True)
            else:
                mean, cov = gp.predict(Xq, return_cov=True)
            np.testing.assert_allclose(mean, g["%s_mean_%d" % (tag, m)], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(cov, g["%s_cov_%d" % (tag, m)], rtol=1e-6, atol=1e-9 * np.abs(cov).max())
    # the Cholesky draw describes the same distribution with other variates, and both modes consume the generator alike
    r1, r2 = np.random.RandomState(9), np.random.RandomState(9)
    a = gp.sample_y(g["Xq_24"], sample_mean=True, n_samples=5, random_state=r1, mvn="reference")
    b = gp.sample_y(g["Xq_24"], sample_mean=True, n_samples=5, random_state=r2, mvn="cholesky")
    assert a.shape == b.shape == (24, 5) and not np.allclose(a, b)
    np.testing.assert_array_equal(r1.get_state()[1], r2.get_state()[1])
    with pytest.raises(ValueError):
        gp.sample_y(g["Xq_6"], mvn="svd")


def test_sample_y_reference_variates_per_hyper_sample_follow_the_reference_loop():
    """sample_mean=False (bask/bayesgpr.py:679-706): all chain-row indices first, then per sample the theta setter and
    sklearn's sample_y on the SAME generator.  The batched device build + per-sample host SVD draw against that loop
    restated with the oracle; theta / alpha_ untouched; generator consumed as in the Cholesky mode."""
    from oracle import gp_oracle as O

    g = load_golden("mvn.npz")
    gp = _mvn_gp(g)
    X, y, Xq = g["X"], g["y"], g["Xq_24"]
    theta0, alpha0 = gp.theta.copy(), gp.alpha_.copy()
    r1 = np.random.RandomState(3)
    got = gp.sample_y(Xq, n_samples=6, random_state=r1, mvn="reference")
    rng = np.random.RandomState(3)
    ind = rng.choice(len(gp.chain_), size=6, replace=True)
    want = np.empty((24, 6))
    for i, j in enumerate(ind):
        want[:, i] = O.sample_y(X, y, np.full(len(y), 1e-10), gp.chain_[j], Xq, 1, rng, noise_zero=True)[:, 0]
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-8)
    np.testing.assert_array_equal(r1.get_state()[1], rng.get_state()[1])
    np.testing.assert_array_equal(gp.theta, theta0)
    np.testing.assert_array_equal(gp.alpha_, alpha0)
    r2 = np.random.RandomState(3)
    other = gp.sample_y(Xq, n_samples=6, random_state=r2, mvn="cholesky")
    assert other.shape == got.shape
    np.testing.assert_array_equal(r1.get_state()[1], r2.get_state()[1])
