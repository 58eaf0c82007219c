"""Generic kernel expression trees (kernels.GramPlan): the reference accepts any kernel (bask/bayesgpr.py:148-159, priors by
recursion over arbitrary Sum / Product trees bask/utils.py:154-179) and evaluates it on the host (sklearn/_gpr.py:582).  For trees
without a canonical device form this build evaluates kernel_(X) with the scikit-learn kernel object as well and hands the matrices
to the device, which does the factorisation, the solves, the log-likelihood, the inverse and the predictive products
(bgp_lml_batch_gram / bgp_posterior_batch_gram / bgp_predict_batch_gram).  Parity against scikit-learn 1.7.2 itself -- the
arithmetic bask/bayesgpr.py:374 reaches -- at the north star's 1e-6."""
import numpy as np
import pytest
from scipy.linalg import cho_solve, cholesky
from sklearn.gaussian_process import GaussianProcessRegressor
from sklearn.gaussian_process import kernels as sk

from conftest import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bask():
    import bayes_skopt_amd as bask

    assert bask._lib.device_count() >= 1
    return bask


def _trees(d):
    return [
        sk.Matern(length_scale=0.4, nu=2.5) * sk.RBF(length_scale=0.7) + sk.WhiteKernel(0.05),
        sk.Matern(length_scale=0.3, nu=2.5) + sk.Matern(length_scale=1.1, nu=1.5) + sk.WhiteKernel(0.02),
        sk.ConstantKernel(0.8) * sk.RationalQuadratic(length_scale=0.5, alpha=1.3) + sk.WhiteKernel(0.03),
        sk.ConstantKernel(1.2) * sk.Matern(length_scale=[0.4] * d, nu=0.7) + sk.WhiteKernel(0.01),
    ]


@pytest.mark.parametrize("n", [100, 300])
def test_lml_of_host_evaluated_kernel_matrices_matches_sklearn(bask, n):
    """bgp_lml_batch_gram: one block column (n = 100) and three (n = 300), scalar and vector alpha, a non-PD item."""
    d = 2
    X, y = synth(n, d, 21)
    rng = np.random.RandomState(3)
    for vec_alpha in (False, True):
        alpha = 1e-10 + (0.01 * rng.uniform(size=n) if vec_alpha else 0.0)
        ad = np.broadcast_to(alpha, (n,)).copy()
        ctx = bask._lib.Context(X, y, ad, max_batch=4)
        for k in _trees(d):
            assert bask.kernels.analyse_kernel(k).generic
            g = GaussianProcessRegressor(kernel=k, optimizer=None, alpha=alpha).fit(X, y)
            T = k.theta + 0.2 * rng.randn(6, len(k.theta))
            want = np.array([g.log_marginal_likelihood(t) for t in T])
            K = np.stack([k.clone_with_theta(t)(X) for t in T])
            got, st = ctx.lml_gram(K, return_status=True)
            assert np.all(st == 0)
            np.testing.assert_allclose(got, want, rtol=1e-6)
        # a matrix that is not positive definite: -inf and dpotrf's info, the others untouched
        K = np.stack([k.clone_with_theta(k.theta)(X) for _ in range(3)])
        K[1, 40, 40] = -1.0
        got, st = ctx.lml_gram(K, return_status=True)
        assert got[1] == -np.inf and st[1] == 41 and st[0] == 0 and st[2] == 0 and got[0] == got[2]
        ctx.close()


def test_posterior_and_predict_from_host_evaluated_matrices(bask):
    n, d, m = 200, 3, 37
    X, y = synth(n, d, 5)
    Xq = np.random.RandomState(6).uniform(size=(m, d))
    ad = np.full(n, 1e-10)
    ctx = bask._lib.Context(X, y, ad, max_batch=4)
    ks = _trees(d)[:3]
    K = np.stack([k(X) for k in ks])
    res = ctx.posterior_gram(K, want_L=True, want_alpha=True, want_K_inv=True)
    assert np.all(res["status"] == 0)
    for b, k in enumerate(ks):
        Kb = K[b] + np.diag(ad)
        L = cholesky(Kb, lower=True)
        a = cho_solve((L, True), y)
        np.testing.assert_allclose(res["L"][b], L, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(res["alpha"][b], a, rtol=1e-6, atol=1e-7 * np.abs(a).max())
        Ki = np.linalg.inv(Kb)
        np.testing.assert_allclose(res["K_inv"][b], Ki, rtol=1e-6, atol=1e-8 * np.abs(Ki).max())
        g = GaussianProcessRegressor(kernel=k, optimizer=None, alpha=1e-10).fit(X, y)
        np.testing.assert_allclose(res["lml"][b], g.log_marginal_likelihood(k.theta), rtol=1e-6)
    Ks = np.stack([k(Xq, X) for k in ks])
    kss = np.stack([k.diag(Xq) for k in ks])
    Kss = np.stack([k(Xq) for k in ks])
    mean, var, cov = ctx.predict_gram(Ks, kss, Kss)
    mean2, var2 = ctx.predict_gram(Ks, kss)
    np.testing.assert_array_equal(mean, mean2)
    np.testing.assert_array_equal(var, var2)
    for b, k in enumerate(ks):
        g = GaussianProcessRegressor(kernel=k, optimizer=None, alpha=1e-10).fit(X, y)
        mu, c = g.predict(Xq, return_cov=True)
        _, sd = g.predict(Xq, return_std=True)
        np.testing.assert_allclose(mean[b], mu, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(np.sqrt(var[b]), sd, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(cov[b], c, rtol=1e-6, atol=1e-8 * np.abs(c).max())
    ctx.close()


@pytest.mark.parametrize("which", ["matern_times_rbf", "matern_plus_matern"])
def test_bayesgpr_fits_trees_the_canonical_analysis_refuses(bask, which):
    """BayesGPR(kernel=Matern(nu=2.5) * RBF() + WhiteKernel()) and Matern() + Matern(): MAP start (L-BFGS-B on the device LML
    with the contraction of scikit-learn's K_gradient), the ensemble MCMC, the geometric median, the posterior, predict and
    the acquisitions -- against scikit-learn's own numbers at the fitted theta."""
    n, d = 140, 2
    X, y = synth(n, d, 8)
    kernel = {
        "matern_times_rbf": sk.Matern(nu=2.5) * sk.RBF() + sk.WhiteKernel(),
        "matern_plus_matern": sk.Matern(length_scale=0.5, nu=2.5) + sk.Matern(length_scale=2.0, nu=1.5),
    }[which]
    gp = bask.BayesGPR(kernel=kernel, normalize_y=False, random_state=4)
    gp.fit(X, y, n_desired_samples=60, n_burnin=3, n_walkers_per_thread=20, progress=False)
    assert gp._generic and gp.chain_.shape == (60, len(gp.theta))
    th = gp.theta
    assert np.all(np.isfinite(th))
    g = GaussianProcessRegressor(kernel=gp.kernel_, optimizer=None, alpha=1e-10).fit(X, y)
    np.testing.assert_allclose(gp.log_marginal_likelihood_value_, g.log_marginal_likelihood(th), rtol=1e-6)
    # every row of the chain: LML through the device against scikit-learn
    rows = gp.chain_[::7]
    np.testing.assert_allclose(gp.log_marginal_likelihood(rows), [g.log_marginal_likelihood(t) for t in rows], rtol=1e-6)
    val, grad = gp.log_marginal_likelihood(th, eval_gradient=True)
    v2, g2 = g.log_marginal_likelihood(th, eval_gradient=True)
    np.testing.assert_allclose(val, v2, rtol=1e-6)
    np.testing.assert_allclose(grad, g2, rtol=1e-5, atol=1e-6)
    Xq = np.random.RandomState(9).uniform(size=(29, d))
    mean, std = gp.predict(Xq, return_std=True)
    mu, sd = g.predict(Xq, return_std=True)
    np.testing.assert_allclose(mean, mu, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(std, sd, rtol=1e-6, atol=1e-8)
    with gp.noise_set_to_zero():
        m0, c0 = gp.predict(Xq, return_cov=True)
        g0 = GaussianProcessRegressor(kernel=gp.kernel_, optimizer=None, alpha=1e-10).fit(X, y)
        g0.L_, g0.alpha_ = g.L_, g.alpha_  # factors keep the noise, the predictive kernel drops it (bask/bayesgpr.py:327-336)
        mu0, cov0 = g0.predict(Xq, return_cov=True)
    np.testing.assert_allclose(m0, mu0, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(c0, cov0, rtol=1e-6, atol=1e-8 * np.abs(cov0).max())
    np.testing.assert_allclose(gp.alpha_, g.alpha_, rtol=1e-6, atol=1e-7 * np.abs(g.alpha_).max())
    np.testing.assert_allclose(gp.L_, g.L_, rtol=1e-8, atol=1e-11)
    # seeded function draws: the reference's own variates (numpy's SVD draw on sklearn's mean / covariance)
    draws = gp.sample_y(Xq, sample_mean=True, n_samples=3, random_state=2)
    want = np.random.RandomState(2).multivariate_normal(mu0, cov0, 3).T
    np.testing.assert_allclose(draws, want, rtol=0, atol=1e-6)
    assert gp.sample_y(Xq, n_samples=2, random_state=1).shape == (29, 2)
    # acquisitions: EI / LCB over hyper-posterior draws against the reference's per-draw loop, PVRS against its (n+1)-Cholesky loop
    A = bask.acquisition
    acqs = [A.ExpectedImprovement(), A.LCB()]
    out = A.evaluate_acquisitions(X=Xq, gpr=gp, acquisition_functions=acqs, random_state=3, n_samples=4)
    idx = np.random.RandomState(3).choice(len(gp.chain_), replace=False, size=4)
    ref = np.zeros((2, len(Xq)))
    backup = gp.theta
    for i in idx:
        gp.theta = gp.chain_[i]
        with gp.noise_set_to_zero():
            mu_i, sd_i = gp.predict(Xq, return_std=True)
        for j, a in enumerate(acqs):
            ref[j] += a(mu_i, sd_i) / 4
    gp.theta = backup
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-12)
    T = Xq[:5]
    covs = gp._pvrs(Xq, T, has_alpha_vec=False)
    k = gp.kernel_
    want = np.empty(len(Xq))
    for i, x in enumerate(Xq):  # bask/acquisition.py:328-338
        Xa = np.vstack([X, x[None, :]])
        L = cholesky(k(Xa), lower=True)
        kt = k(Xa, T)
        want[i] = np.sum(kt * cho_solve((L, True), kt))
    np.testing.assert_allclose(covs, want, rtol=1e-6)
    pv = A.evaluate_acquisitions(X=Xq, gpr=gp, acquisition_functions=[A.PVRS(), A.ThompsonSampling()], random_state=1,
                                 n_samples=1)
    assert pv.shape == (2, 29) and np.all(np.isfinite(pv))


@pytest.mark.parametrize("which", ["matern_times_rbf", "matern_plus_matern", "matern_nu_0.7", "warped"])
def test_prediction_gradients_on_generic_trees(bask, which):
    """``predict(return_mean_grad=True, return_std_grad=True)`` (skopt's single-point gradients, forwarded by
    ``bask/bayesgpr.py:622-635``) on trees without a canonical device form: ``kernels.gradient_x`` + device ``alpha_`` /
    ``K_inv_`` against central differences of the device predict."""
    n, d = 120, 2
    X, y = synth(n, d, 8)
    kernel = {
        "matern_times_rbf": sk.Matern(nu=2.5) * sk.RBF() + sk.WhiteKernel(),
        "matern_plus_matern": sk.Matern(length_scale=0.5, nu=2.5) + sk.Matern(length_scale=2.0, nu=1.5),
        "matern_nu_0.7": sk.ConstantKernel(1.2) * sk.Matern(length_scale=[0.4] * d, nu=0.7),
        "warped": sk.Matern(length_scale=0.5, nu=2.5) + sk.Matern(length_scale=2.0, nu=1.5),
    }[which]
    gp = bask.BayesGPR(kernel=kernel, normalize_y=True, random_state=4, warp_inputs=(which == "warped"))
    gp.fit(X, y, n_desired_samples=40, n_burnin=2, n_walkers_per_thread=20, progress=False)
    assert gp._generic
    for x in np.random.RandomState(2).uniform(0.1, 0.9, size=(3, d)):
        m, sd, gm, gs = gp.predict(x[None, :], return_std=True, return_mean_grad=True, return_std_grad=True)
        m2, gm2 = gp.predict(x[None, :], return_mean_grad=True)
        np.testing.assert_array_equal(gm, gm2)
        if which == "warped":
            # the derivative is with respect to the WARPED coordinates, as in the reference (the kernel never sees anything else):
            # difference the predict in the warped coordinate through the inverse warp
            xw = gp.warp(x[None, :])[0]
            to_x = lambda u: gp.unwarp(u[None, :])  # noqa: E731
        else:
            xw, to_x = x, (lambda u: u[None, :])
        h = 1e-5
        fd_m, fd_s = np.zeros(d), np.zeros(d)
        for j in range(d):
            e = np.zeros(d)
            e[j] = h
            mp, sp = gp.predict(to_x(xw + e), return_std=True)
            mm, sm = gp.predict(to_x(xw - e), return_std=True)
            fd_m[j], fd_s[j] = (mp[0] - mm[0]) / (2 * h), (sp[0] - sm[0]) / (2 * h)
        np.testing.assert_allclose(gm, fd_m, rtol=2e-5, atol=1e-7 * max(1.0, np.abs(fd_m).max()))
        np.testing.assert_allclose(gs, fd_s, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(fd_s).max()))


def test_a_gradient_at_another_theta_does_not_leave_a_stale_resident_posterior(bask):
    """``log_marginal_likelihood(theta', eval_gradient=True)`` on a generic tree overwrites the device-resident K^-1 / alpha; the
    next predict must rebuild the posterior of ``theta`` (advisor, round 5)."""
    n, d = 120, 2
    X, y = synth(n, d, 8)
    gp = bask.BayesGPR(kernel=sk.Matern(length_scale=0.5, nu=2.5) + sk.Matern(length_scale=2.0, nu=1.5), random_state=4)
    gp.fit(X, y, n_desired_samples=40, n_burnin=2, n_walkers_per_thread=20, progress=False)
    Xq = np.random.RandomState(9).uniform(size=(17, d))
    before = gp.predict(Xq, return_std=True)
    gp.log_marginal_likelihood(gp.theta + 0.7, eval_gradient=True)
    after = gp.predict(Xq, return_std=True)
    np.testing.assert_array_equal(before[0], after[0])
    np.testing.assert_array_equal(before[1], after[1])
    g = GaussianProcessRegressor(kernel=gp.kernel_, optimizer=None, alpha=1e-10).fit(X, y)
    mu, sd_ = g.predict(Xq, return_std=True)
    np.testing.assert_allclose(after[0], mu, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(after[1], sd_, rtol=1e-6, atol=1e-8)


def test_optimizer_diagnostics_on_a_generic_tree(bask):
    """``expected_minimum`` (L-BFGS on the predictive mean with its gradient) and through it ``probability_of_optimality`` /
    ``expected_optimality_gap`` / ``optimum_intervals`` (``bask/optimizer.py:447-689``) on a kernel tree the canonical analysis
    refuses: they need the prediction gradients above."""
    rng = np.random.RandomState(0)
    opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * 2, n_points=200, n_initial_points=5, acq_func="ei", random_state=0,
                         gp_kernel=sk.ConstantKernel(1.0, (0.1, 10.0)) * sk.Matern(0.5, (0.05, 5.0), nu=2.5)
                         * sk.RBF(1.0, (0.05, 5.0)))
    for _ in range(14):
        x = opt.ask()
        opt.tell(x, float((x[0] - 0.3) ** 2 + (x[1] - 0.6) ** 2 + 0.01 * rng.randn()), gp_samples=100, gp_burnin=2, n_samples=0)
    assert opt.gp._generic
    from bayes_skopt_amd.utils import expected_minimum
    res = opt._result()
    x_opt, f_opt = expected_minimum(res, n_random_starts=20, random_state=0)[:2]
    assert all(0.0 <= v <= 1.0 for v in x_opt) and np.isfinite(f_opt)
    assert f_opt <= opt.gp.predict(np.array([[0.95, 0.05]]))[0]  # (the far corner of the bowl is predicted worse)
    p = opt.probability_of_optimality(threshold=[0.5, 2.0], n_space_samples=100, n_gp_samples=50, n_random_starts=10, random_state=0)
    assert len(p) == 2 and 0.0 <= p[0] <= p[1] <= 1.0
    gap = opt.expected_optimality_gap(n_probabilities=10, n_space_samples=100, n_gp_samples=50, n_random_starts=10, random_state=0)
    assert np.isfinite(gap) and gap >= 0.0
    iv = opt.optimum_intervals(opt_samples=50, space_samples=100, random_state=0)
    assert len(iv) == 2


def test_optimizer_runs_on_a_generic_tree(bask):
    rng = np.random.RandomState(0)
    opt = bask.Optimizer(dimensions=[(0.0, 1.0)] * 2, n_points=150, n_initial_points=5, acq_func="pvrs", random_state=0,
                         gp_kernel=sk.ConstantKernel(1.0, (0.1, 10.0)) * sk.Matern(0.5, (0.05, 5.0), nu=2.5)
                         * sk.RBF(1.0, (0.05, 5.0)))
    for _ in range(8):
        x = opt.ask()
        opt.tell(x, float(np.sin(4 * x[0]) + x[1] ** 2 + 0.01 * rng.randn()), gp_samples=100, gp_burnin=2, n_samples=0)
    assert opt.gp._generic and opt.gp.chain_.shape[0] == 100
    assert all(0.0 <= v <= 1.0 for v in opt.ask())
    # a kernel the reference's own guess_priors refuses (bask/utils.py:178-179) is refused alike -- with explicit priors it runs
    with pytest.raises(NotImplementedError):
        bask.guess_priors(sk.RationalQuadratic())


def test_generic_tree_with_input_warping(bask):
    """warp_inputs=True on a tree without a canonical form: every walker's kernel matrix is evaluated on the host from ITS warped
    inputs (the device's Beta-CDF warp), the chain carries the 2d warp parameters, predictions warp the query points; the LML of a
    chain row equals scikit-learn's on the row's warped design."""
    from oracle import gp_oracle as O

    n, d = 90, 1
    rng = np.random.RandomState(3)
    X = rng.uniform(size=(n, d))
    y = np.sin(6.0 * X[:, 0] ** 2) + 0.05 * rng.randn(n)
    kernel = sk.Matern(length_scale=0.4, nu=2.5) + sk.Matern(length_scale=1.5, nu=1.5)
    gp = bask.BayesGPR(kernel=kernel, random_state=2, warp_inputs=True)
    gp.fit(X, y, n_desired_samples=40, n_burnin=2, n_walkers_per_thread=20, progress=False)
    assert gp._generic and gp.chain_.shape == (40, 3 + 2 * d)
    row = gp.chain_[7]
    th, w = row[:3], row[3:]
    Xw = O.warp_inputs(X, w)
    g = GaussianProcessRegressor(kernel=gp.kernel_, optimizer=None, alpha=1e-10).fit(Xw, gp.y_train_)
    got = gp._gram_lml(th[None, :], w[None, :])[0]
    np.testing.assert_allclose(got, g.log_marginal_likelihood(th), rtol=1e-6)
    Xq = np.linspace(0.02, 0.98, 15)[:, None]
    mean, std = gp.predict(Xq, return_std=True)
    wm = np.concatenate([gp.warp_alphas_, gp.warp_betas_])
    gm = GaussianProcessRegressor(kernel=gp.kernel_, optimizer=None, alpha=1e-10).fit(O.warp_inputs(X, wm), gp.y_train_)
    mu, sd = gm.predict(O.warp_inputs(Xq, wm), return_std=True)
    np.testing.assert_allclose(mean, mu * gp.y_train_std_ + gp.y_train_mean_, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(std, sd * gp.y_train_std_, rtol=1e-6, atol=1e-8)
    acq = bask.acquisition.evaluate_acquisitions(Xq, gp, [bask.acquisition.ExpectedImprovement()], n_samples=3, random_state=0)
    assert acq.shape == (1, 15) and np.all(np.isfinite(acq))
