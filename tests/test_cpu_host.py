"""CPU-only tests: host logic (kernel analysis, priors, sampler, geometric median, init sequences),
the C-ABI library (loads, exports every declared symbol, fails loudly without a device) and the
product's isolation from the oracle."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def bask():
    import bayes_skopt_amd as bask

    return bask


def test_library_exports_every_declared_symbol(bask):
    """include/bgp.h is the contract: every function it declares must be exported by libbgp.so and
    bound by the ctypes layer."""
    hdr = open(os.path.join(ROOT, "include", "bgp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bgp_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = bask._lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(bask._lib.SIGNATURES), declared ^ set(bask._lib.SIGNATURES)
    assert b"gfx950" in lib.bgp_version()


def test_no_device_fails_loudly(bask):
    if bask._lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(bask._lib.BgpError):
        bask._lib.Context(np.zeros((4, 2)), np.zeros(4), 1e-10)
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1]), random_state=0)
    with pytest.raises(bask._lib.BgpError):
        gp.fit(np.random.rand(8, 2), np.random.rand(8), progress=False)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "bayes-skopt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/gp_oracle.py (", ""), f


def test_kernel_analysis_and_canonical_mapping(bask):
    from sklearn.gaussian_process import kernels as sk

    K = bask.kernels
    k = sk.ConstantKernel(1.0, (0.1, 2.0)) * sk.Matern([0.3, 0.4], (0.2, 0.5), nu=2.5) + sk.WhiteKernel(0.01)
    pl = K.analyse_kernel(k)
    assert (pl.form, pl.stationary, pl.n_theta) == ("product", "matern52", 4)
    np.testing.assert_allclose(pl.canonical(k.theta, 2), [np.log([1.0, 0.3, 0.4, 0.01])])
    # isotropic length scale is replicated, fixed constant dropped from theta, operand order free
    k2 = sk.WhiteKernel(0.1) + sk.RBF(0.5) * sk.ConstantKernel(2.0, "fixed")
    pl2 = K.analyse_kernel(k2)
    assert (pl2.form, pl2.stationary, pl2.n_theta) == ("product", "rbf", 2)
    np.testing.assert_allclose(pl2.canonical(k2.theta, 3), [np.log([2.0, 0.5, 0.5, 0.5, 0.1])])
    g = np.arange(5.0)[None, :]
    np.testing.assert_allclose(pl2.grad_to_theta(g, 3), [[4.0, 1.0 + 2.0 + 3.0]])
    # the notebook's sum form, no white kernel, zeroed white kernel
    k3 = sk.ConstantKernel(1.0) + sk.Matern(0.3, nu=1.5)
    pl3 = K.analyse_kernel(k3)
    assert (pl3.form, pl3.stationary) == ("sum", "matern32")
    assert pl3.canonical(k3.theta, 2)[0, -1] == -np.inf
    k4 = k.clone_with_theta(k.theta)
    k4.set_params(k2=sk.WhiteKernel(noise_level=0.0))
    with np.errstate(divide="ignore"):
        assert K.analyse_kernel(k4).canonical(k4.theta, 2)[0, -1] == -np.inf
    # trees without a canonical device form: a GramPlan (host-evaluated kernel matrices, device arithmetic); strict refuses
    for other in (sk.DotProduct(), sk.RBF(1.0) * sk.RBF(2.0), sk.Matern(nu=0.7), sk.RBF() + sk.RBF(),
                  sk.Matern(nu=2.5) * sk.RBF() + sk.WhiteKernel(), sk.RationalQuadratic()):
        plan = K.analyse_kernel(other)
        assert isinstance(plan, K.GramPlan) and plan.generic and plan.n_theta == len(other.theta)
        with pytest.raises(NotImplementedError):
            plan.canonical(other.theta, 2)
        with pytest.raises(NotImplementedError):
            K.analyse_kernel(other, strict=True)
    assert not pl.generic
    with pytest.raises(TypeError):
        K.analyse_kernel("rbf")
    assert K.param_for_white_kernel_in_sum(k) == (True, "k2")
    assert K.param_for_white_kernel_in_sum(k3)[0] is False


def test_default_kernel_and_priors_known_answers(bask):
    """reference tests/test_utils.py:15-40"""
    from sklearn.gaussian_process import kernels as sk

    assert len(bask.construct_default_kernel([0, 1]).theta) == 3
    g = load_golden("reference_tier1.npz")
    dk = bask.construct_default_kernel([0, 1, 2])
    np.testing.assert_allclose(dk.theta, g["default_kernel_theta"])
    np.testing.assert_allclose(dk.bounds, g["default_kernel_bounds"])
    kernel = sk.ConstantKernel(1.0, (0.1, 2.0)) * sk.Matern([0.3, 0.3], (0.2, 0.5), nu=2.5) + sk.WhiteKernel()
    pri = bask.guess_priors(kernel)
    assert len(pri) == 4
    assert pri[1](-0.9) == pytest.approx(-0.02116327824572739, abs=1e-13)
    assert pri[0](-0.9) == pytest.approx(-2.112906921232193, abs=1e-13)
    t = g["prior_grid"]
    np.testing.assert_allclose(pri[0](t), g["prior_variance"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(pri[3](t), g["prior_noise"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(pri[2](t), g["prior_lengthscale"], rtol=1e-12, atol=1e-12)
    # fixed hyper-parameters get no prior; nested kernels are walked recursively
    nested = sk.ConstantKernel(1.0, "fixed") * sk.RBF([1.0, 2.0]) + sk.WhiteKernel(1.0, "fixed")
    assert len(bask.guess_priors(nested)) == 2
    with pytest.raises(NotImplementedError):
        bask.guess_priors(sk.DotProduct())
    rf = bask.priors.make_roundflat()
    np.testing.assert_allclose(rf(g["roundflat_x"]), g["roundflat_val"], rtol=1e-12)
    rf2 = bask.priors.make_roundflat(0.2, 0.9, 3.0, 4.0)
    np.testing.assert_allclose(rf2(g["roundflat_x"]), g["roundflat2_val"], rtol=1e-12)


def test_roundflat_integrates_to_one(bask):
    """reference tests/test_priors.py:8-11"""
    from scipy.integrate import quad

    prior = bask.priors.make_roundflat()
    assert quad(lambda x: np.exp(prior(x)), 0.0, 10.0)[0] == pytest.approx(1.0, abs=1e-8)


def test_geometric_median_golden(bask):
    g = load_golden("reference_tier1.npz")
    for i in range(3):
        np.testing.assert_allclose(bask.geometric_median(g[f"gm{i}_chain"]), g[f"gm{i}_median"], rtol=1e-12, atol=1e-13)


def test_validate_zeroone(bask):
    """reference tests/test_utils.py:43-49"""
    bask.utils.validate_zeroone(np.linspace(0, 1, 5))
    with pytest.raises(ValueError):
        bask.utils.validate_zeroone(np.array([0.5, 1.1]))
    with pytest.raises(ValueError):
        bask.utils.validate_zeroone([-0.1, 0.2])


def test_top_level_names_are_the_reference_packages(bask):
    """``import bayes_skopt_amd as bask`` offers every name ``bask/__init__.py:19-35`` exports (the acquisition classes are
    top-level names there), the helpers other modules of the reference import (``bask/utils.py:198`` ``get_progress_bar``,
    ``bask/init.py:90`` ``phi``), and the constructors / methods take the reference's arguments in the reference's order."""
    import inspect

    for name in ("BayesGPR", "Optimizer", "BayesSearchCV", "guess_priors", "evaluate_acquisitions", "ExpectedImprovement",
                 "TopTwoEI", "Expectation", "LCB", "MaxValueSearch", "r2_sequence", "sb_sequence", "ThompsonSampling",
                 "VarianceReduction", "PVRS"):
        assert name in bask.__all__ and getattr(bask, name) is not None, name
    with bask.utils.get_progress_bar(False, 3) as bar:
        bar.update(1)
    assert bask.init.phi(1) == pytest.approx((1 + 5 ** 0.5) / 2) and bask.init.phi(2) ** 3 == pytest.approx(bask.init.phi(2) + 1)
    x = bask.init.phi(5, n_iter=60)
    assert x ** 6 == pytest.approx(x + 1)

    def leading(f, names):
        got = list(inspect.signature(f).parameters)[: len(names)]
        assert got == names, (f.__qualname__, got)

    leading(bask.BayesGPR.__init__, ["self", "kernel", "alpha", "optimizer", "n_restarts_optimizer", "normalize_y", "warp_inputs",
                                     "copy_X_train", "random_state", "noise"])
    leading(bask.BayesGPR.fit, ["self", "X", "y", "noise_vector", "n_threads", "n_desired_samples", "n_burnin",
                                "n_walkers_per_thread", "progress", "priors", "warp_priors", "position"])
    leading(bask.BayesGPR.sample, ["self", "X", "y", "noise_vector", "n_threads", "n_desired_samples", "n_burnin", "n_thin",
                                   "n_walkers_per_thread", "progress", "priors", "warp_priors", "position", "add"])
    leading(bask.BayesGPR.predict, ["self", "X", "return_std", "return_cov", "return_mean_grad", "return_std_grad"])
    leading(bask.BayesGPR.sample_y, ["self", "X", "sample_mean", "noise", "n_samples", "random_state"])
    leading(bask.Optimizer.tell, ["self", "x", "y", "noise_vector", "fit", "replace", "n_samples", "gp_samples", "gp_burnin",
                                  "progress"])
    leading(bask.evaluate_acquisitions, ["X", "gpr", "acquisition_functions", "n_samples", "progress", "random_state"])


def test_init_sequences(bask):
    """reference tests/test_init.py:6-21"""
    assert bask.init.sb_sequence(3, 2, random_state=0).shape == (3, 2)
    ex = [[0.1, 0.2], [0.5, 0.5]]
    out = bask.init.sb_sequence(4, 2, existing_points=ex, random_state=0)
    assert out.shape == (4, 2) and np.allclose(out[:2], ex)
    with pytest.raises(ValueError):
        bask.init.sb_sequence(2, 2, existing_points=ex)
    z = bask.r2_sequence(5, 3)
    assert z.shape == (5, 3) and np.all((z >= 0) & (z < 1))


def test_sampler_matches_oracle_sampler_bitwise(bask):
    """Same RandomState stream + same log-prob => identical trajectory: the batched host sampler
    consumes the RNG exactly like the per-walker restatement of emcee's loop (oracle)."""
    from oracle import gp_oracle as O

    p = 3
    rng = np.random.RandomState(0)
    A = rng.randn(p, p)
    icov = np.linalg.inv(A @ A.T + np.eye(p))
    mu = np.array([1.0, -2.0, 0.5])

    def lp_one(x):
        return -0.5 * (x - mu) @ icov @ (x - mu)

    def lp_vec(Xb):
        return np.array([lp_one(x) for x in Xb])

    p0 = mu + 1e-2 * rng.randn(12, p)
    chain_o, lps_o, pos_o, lpf_o, nacc = O.stretch_move_sampler(lp_one, p0, 50, np.random.RandomState(5))
    s = bask.sampler.EnsembleSampler(12, p, lp_vec)
    s.random_state = np.random.RandomState(5).get_state()
    st = s.run_mcmc(p0, 50)
    np.testing.assert_array_equal(s.get_chain(), chain_o)
    np.testing.assert_array_equal(s.get_log_prob(), lps_o)
    np.testing.assert_array_equal(st.coords, pos_o)
    np.testing.assert_array_equal(s.naccepted, nacc)
    flat = s.get_chain(flat=True, discard=10, thin=2)
    assert flat.shape == (20 * 12, p)
    np.testing.assert_array_equal(flat[:12], chain_o[11])  # step-major, emcee's discard+thin-1 start
    coords, log_prob, rstate = st
    assert coords.shape == (12, p) and log_prob.shape == (12,)

    # a log_prob_fn with begin()/finish() (BayesGPR's device batch) gets the block asynchronously: the sampler draws the
    # accept uniforms and the NEXT half-step's stretch factors / partners in between -- same stream order, same chain,
    # and the generator ends in the same state (nothing drawn past the last half-step)
    class Split:
        calls = 0

        def __call__(self, Xb):
            return lp_vec(Xb)

        def begin(self, Xb):
            Split.calls += 1
            return np.array(Xb)

        def finish(self, token):
            return lp_vec(token)

    s2 = bask.sampler.EnsembleSampler(12, p, Split())
    s2.random_state = np.random.RandomState(5).get_state()
    st2 = s2.run_mcmc(p0, 30)
    st2 = s2.run_mcmc(st2.coords, 20, log_prob0=st2.log_prob)  # resumed like BayesGPR.sample with pos_
    assert Split.calls == 100
    np.testing.assert_array_equal(s2.get_chain(), chain_o)
    np.testing.assert_array_equal(s2.naccepted, nacc)
    assert s2._random.rand() == s._random.rand()


def test_resident_plan_replays_the_host_loop_bitwise(bask):
    """The device-resident run (bgp_mcmc_run) gets every random number of the run as a plan drawn up front
    (EnsembleSampler._run_resident).  A numpy replay of what mcmc_step_kernel does with that plan -- accept half-step h - 1,
    propose half-step h, ensemble into the chain after the second half of a step -- must give the host loop's chain, accept
    counts and generator end state bit for bit: the plan IS emcee's stream, in emcee's order."""
    p = 3
    rng = np.random.RandomState(0)
    A = rng.randn(p, p)
    icov = np.linalg.inv(A @ A.T + np.eye(p))
    mu = np.array([1.0, -2.0, 0.5])

    def lp_vec(Xb):
        return np.array([-0.5 * (x - mu) @ icov @ (x - mu) for x in Xb])

    class Resident:
        asked = 0

        def __call__(self, Xb):
            return lp_vec(Xb)

        def resident(self, n_walkers, n_dim):
            Resident.asked += 1

            class Run:
                def begin(self, coords, log_prob, nsteps):
                    self.coords, self.log_prob = np.array(coords), np.array(log_prob)
                    self.chain, self.lps = np.empty((nsteps,) + self.coords.shape), np.empty((nsteps, len(self.coords)))
                    self.nacc, self.h = np.zeros(len(self.coords), dtype=np.int64), 0
                    self.segments = 0

                def steps(self, plan):
                    movers, partners, zz, factors, logu = plan
                    self.segments += 1
                    for r in range(movers.shape[0]):
                        s, c = self.coords[movers[r]], self.coords[partners[r]]
                        q = c - (c - s) * zz[r][:, None]
                        new_lp = lp_vec(q)
                        acc = factors[r] + new_lp - self.log_prob[movers[r]] > logu[r]
                        idx = movers[r][acc]
                        self.coords[idx], self.log_prob[idx] = q[acc], new_lp[acc]
                        self.nacc[idx] += 1
                        if self.h & 1:
                            self.chain[self.h // 2], self.lps[self.h // 2] = self.coords, self.log_prob
                        self.h += 1

                def end(self):
                    assert self.h == 2 * len(self.chain) and self.segments >= 3  # (handed over in growing segments)
                    return self.chain, self.lps, self.coords, self.log_prob, self.nacc, np.zeros(2, dtype=np.int32)

                def abandon(self):
                    pass

            return Run()

    p0 = mu + 1e-2 * rng.randn(12, p)
    ref = bask.sampler.EnsembleSampler(12, p, lp_vec)
    ref.random_state = np.random.RandomState(5).get_state()
    st0 = ref.run_mcmc(p0, 30)
    st0 = ref.run_mcmc(st0.coords, 20, log_prob0=st0.log_prob)
    s = bask.sampler.EnsembleSampler(12, p, Resident())
    s.random_state = np.random.RandomState(5).get_state()
    st = s.run_mcmc(p0, 30)
    st = s.run_mcmc(st.coords, 20, log_prob0=st.log_prob)
    assert Resident.asked == 2 and s.resident_runs == 2 and getattr(ref, "resident_runs", 0) == 0
    np.testing.assert_array_equal(s.get_chain(), ref.get_chain())
    np.testing.assert_array_equal(s.get_log_prob(), ref.get_log_prob())
    np.testing.assert_array_equal(s.naccepted, ref.naccepted)
    np.testing.assert_array_equal(st.coords, st0.coords)
    assert s.iteration == ref.iteration == 50 and s.n_log_prob_evals == ref.n_log_prob_evals
    assert s._random.rand() == ref._random.rand()
    # a progress bar does not keep the run on the host (fit()'s default is progress=True) ...
    s3 = bask.sampler.EnsembleSampler(12, p, Resident())
    s3.random_state = np.random.RandomState(5).get_state()
    s3.run_mcmc(p0, 10, progress=True)
    assert s3.resident_runs == 1
    np.testing.assert_array_equal(s3.get_chain(), ref.get_chain()[:10])


def test_a_declined_resident_run_says_why_once(bask, capsys):
    """A log_prob_fn whose resident() declines keeps the host loop, and the reason it gives goes to stderr once per process."""
    class Declines:
        resident_reason = "a test reason %d" % id(object())

        def __call__(self, Xb):
            return -0.5 * (Xb * Xb).sum(axis=1)

        def resident(self, n_walkers, n_dim):
            return None

    p0 = 1e-2 * np.random.RandomState(0).randn(12, 3)
    for _ in range(3):
        s = bask.sampler.EnsembleSampler(12, 3, Declines())
        s.run_mcmc(p0, 4)
        assert getattr(s, "resident_runs", 0) == 0 and s.get_chain().shape == (4, 12, 3)
    err = capsys.readouterr().err
    assert err.count(Declines.resident_reason) == 1 and "driven from the host" in err


def test_sampler_preconditions(bask):
    s = bask.sampler.EnsembleSampler(4, 3, lambda X: np.zeros(len(X)))
    with pytest.raises(RuntimeError):
        s.run_mcmc(np.random.rand(4, 3), 1)  # fewer walkers than 2*ndim
    s = bask.sampler.EnsembleSampler(8, 2, lambda X: np.zeros(len(X)))
    with pytest.raises(ValueError):
        s.run_mcmc(np.ones((8, 2)), 1)  # degenerate ensemble
    s = bask.sampler.EnsembleSampler(8, 2, lambda X: np.full(len(X), np.nan))
    with pytest.raises(ValueError):
        s.run_mcmc(np.random.rand(8, 2), 1)  # NaN log-prob
    s = bask.sampler.EnsembleSampler(8, 2, lambda X: np.zeros(len(X)))
    bad = np.random.rand(8, 2)
    bad[0, 0] = np.inf
    with pytest.raises(ValueError):
        s.run_mcmc(bad, 1, skip_initial_state_check=True)


def test_bayesgpr_host_semantics_without_device(bask):
    """Argument handling that does not need the GPU."""
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1]), random_state=0)
    assert gp.theta is None and gp.chain_ is None and gp.pos_ is None
    with pytest.raises(ValueError):
        gp.sample()
    gp._apply_noise_vector(3, np.array([1.0, 2.0]))
    np.testing.assert_allclose(gp.alpha, [1.0 + 1e-10, 2.0 + 1e-10, 1e-10])
    gp._apply_noise_vector(3, np.array([0.5, 0.5, 0.5]))  # re-applied on the ORIGINAL scalar alpha
    np.testing.assert_allclose(gp.alpha, 0.5 + 1e-10)
    gw = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1]), warp_inputs=True)
    assert gw.warp_inputs and gw.warp(np.ones((2, 2))).shape == (2, 2)  # identity before the first fit
    from bayes_skopt_amd.bayesgpr import _eval_priors

    Theta = np.random.RandomState(0).randn(5, 2)
    vec = [lambda t: -0.5 * t**2, lambda t: float(-abs(t))]  # second one only accepts scalars
    np.testing.assert_allclose(_eval_priors(vec, Theta), -0.5 * Theta[:, 0] ** 2 - np.abs(Theta[:, 1]))
    np.testing.assert_allclose(_eval_priors(lambda row: row.sum(), Theta), Theta.sum(axis=1))
    with pytest.raises(ValueError):
        _eval_priors(vec[:1], Theta)


def test_hdi_unimodal_is_narrowest_interval_and_multimodal_splits():
    """utils.hdi restates arviz.hdi (bask/optimizer.py:684): the unimodal estimator is the narrowest interval
    holding hdi_prob of the sorted sample (checked against brute force), the multimodal one returns one interval
    per mode."""
    from bayes_skopt_amd.utils import hdi

    rng = np.random.RandomState(3)
    x = rng.gamma(2.0, size=501)
    lo, hi = hdi(x, 0.9)
    xs = np.sort(x)
    k = int(np.floor(0.9 * len(xs)))
    best = min((xs[i + k] - xs[i], xs[i], xs[i + k]) for i in range(len(xs) - k))
    assert (lo, hi) == (best[1], best[2])
    assert np.mean((x >= lo) & (x <= hi)) >= 0.9
    two = np.concatenate([0.25 + 0.03 * rng.randn(600), 0.75 + 0.03 * rng.randn(600)])
    iv = hdi(two, 0.9, multimodal=True)
    assert iv.shape == (2, 2)
    assert iv[0, 0] < 0.25 < iv[0, 1] < 0.5 < iv[1, 0] < 0.75 < iv[1, 1]
    assert hdi(two, 0.9, multimodal=False).shape == (2,)
    assert hdi(np.full(10, 0.3), 0.9, multimodal=True).tolist() == [[0.3, 0.3]]


def test_expected_minimum_on_a_known_surrogate():
    """utils.expected_minimum (skopt.utils.expected_minimum, called at bask/optimizer.py:497): bounded L-BFGS-B on
    the surrogate mean from the incumbent + random starts, in the ORIGINAL space."""
    import bayes_skopt_amd as bask
    from bayes_skopt_amd.space import Space, create_result
    from bayes_skopt_amd.utils import expected_minimum

    space = Space([(-2.0, 2.0), (0.0, 10.0)])

    class Model:  # mean over the unit cube with a unique minimum at the image of (0.5, 7.0), a local one elsewhere
        calls = 0

        def predict(self, Xt):
            Model.calls += 1
            Xt = np.asarray(Xt)
            assert Xt.min() >= -1e-6 and Xt.max() <= 1 + 1e-6  # transformed coordinates, inside the bounds
            a, b = Xt[:, 0] - 0.625, Xt[:, 1] - 0.7
            return 3.0 * a * a + b * b - 0.3 * np.exp(-200.0 * ((Xt[:, 0] - 0.1) ** 2 + (Xt[:, 1] - 0.1) ** 2))

    res = create_result([[-1.6, 1.0]], [0.0], space, models=[Model()])
    x, f = expected_minimum(res, n_random_starts=15, random_state=0)
    np.testing.assert_allclose(x, [0.5, 7.0], atol=1e-4)
    assert abs(f) < 1e-8
    # a start on the upper bound must difference backwards (never leaves the box)
    res2 = create_result([[2.0, 10.0]], [0.0], space, models=[Model()])
    x2, _ = expected_minimum(res2, n_random_starts=0)
    np.testing.assert_allclose(x2, [0.5, 7.0], atol=1e-4)
    with pytest.raises(ValueError):
        expected_minimum(create_result([["a"]], [0.0], Space([["a", "b"]]), models=[Model()]))


def test_searchcv_initial_design_only_runs_on_cpu():
    """BayesSearchCV (bask/searchcv.py) with fewer iterations than initial points never fits the GP, so the
    scikit-learn plumbing (clone, cv_results_, refit, point <-> dict mapping in sorted key order) is testable
    without a device."""
    from sklearn.base import clone
    from sklearn.datasets import load_iris
    from sklearn.model_selection import train_test_split
    from sklearn.svm import SVC

    import bayes_skopt_amd as bask
    from bayes_skopt_amd.space import Categorical, Integer, Real

    X, y = load_iris(return_X_y=True)
    Xtr, Xte, ytr, yte = train_test_split(X, y, train_size=0.75, random_state=0)
    spaces = {
        "C": Real(1e-6, 1e6, prior="log-uniform"),
        "gamma": Real(1e-6, 1e1, prior="log-uniform"),
        "degree": Integer(1, 8),
        "kernel": Categorical(["linear", "poly", "rbf"]),
    }
    opt = bask.BayesSearchCV(SVC(), spaces, n_iter=6, cv=3, random_state=0)
    assert clone(opt).get_params()["n_iter"] == 6 and opt.total_iterations == 6
    opt.fit(Xtr, ytr)
    assert len(opt.cv_results_["params"]) == 6
    assert set(opt.best_params_) == set(spaces)
    assert opt.best_params_["kernel"] in ("linear", "poly", "rbf") and 1 <= opt.best_params_["degree"] <= 8
    assert opt.best_score_ == max(opt.cv_results_["mean_test_score"])
    assert opt.score(Xte, yte) > 0.6
    assert len(opt.optimizer_results_) == 1 and len(opt.optimizer_results_[0].x_iters) == 6
    assert opt.optimizer_kwargs_["acq_func"] == "pvrs"  # bask/searchcv.py:289-290
    # list-of-(dict, n_iter) form
    two = bask.BayesSearchCV(SVC(), [({"C": Real(1e-3, 1e3, prior="log-uniform")}, 2), ({"gamma": Real(1e-4, 1.0)}, 3)],
                             cv=3, random_state=1)
    assert two.total_iterations == 5
    two.fit(Xtr, ytr)
    assert len(two.cv_results_["params"]) == 5 and len(two.optimizer_results_) == 2
    with pytest.raises(ValueError):
        bask.BayesSearchCV(SVC(), [({"C": Real(1e-3, 1e3)}, 0)]).fit(Xtr, ytr)


def test_bayesgpr_is_an_sklearn_estimator_without_a_device():
    """get_params / set_params / clone / pickle of an UNFITTED BayesGPR need no device (the reference's class is
    an sklearn estimator through skopt's GaussianProcessRegressor)."""
    import copy
    import pickle

    from sklearn.base import clone

    import bayes_skopt_amd as bask
    from bayes_skopt_amd.kernels import ConstantKernel, Matern

    gp = bask.BayesGPR(kernel=ConstantKernel(1.0) * Matern(0.3, nu=2.5), normalize_y=True, random_state=1, warp_inputs=True)
    params = gp.get_params(deep=False)
    assert params["normalize_y"] is True and params["warp_inputs"] is True and params["noise"] == "gaussian"
    g2 = clone(gp)
    assert g2 is not gp and g2.normalize_y is True and g2.chain_ is None
    g2.set_params(normalize_y=False, kernel__k2__length_scale=0.5)
    assert g2.normalize_y is False and g2.kernel.k2.length_scale == 0.5 and gp.kernel.k2.length_scale == 0.3
    for other in (pickle.loads(pickle.dumps(gp)), copy.deepcopy(gp)):
        assert other.get_params(deep=False).keys() == params.keys() and other._ctx is None


def test_closed_form_acquisitions_match_the_reference_classes():
    """The product's own acquisition classes (host numpy closed forms on (mu, std)) against the outputs of the
    reference's ``bask.acquisition`` classes on the same inputs (``tests/golden/reference_tier1.npz``, generated by
    importing the reference): EI (default and explicit optimum), LCB (default and alpha = 3), mean and TopTwoEI
    (``bask/acquisition.py:154-216``)."""
    from bayes_skopt_amd import acquisition as acq

    g = load_golden("reference_tier1.npz")
    mu, std = g["acq_mu"], g["acq_std"]
    np.testing.assert_allclose(acq.ExpectedImprovement()(mu, std), g["acq_ei"], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(acq.ExpectedImprovement()(mu, std, y_opt=-0.3), g["acq_ei_yopt"], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(acq.LCB()(mu, std), g["acq_lcb"], rtol=1e-13)
    np.testing.assert_allclose(acq.LCB()(mu, std, alpha=3.0), g["acq_lcb3"], rtol=1e-13)
    np.testing.assert_allclose(acq.Expectation()(mu, std), g["acq_mean"], rtol=1e-13)
    np.testing.assert_allclose(acq.TopTwoEI()(mu, std), g["acq_ttei"], rtol=1e-13, atol=1e-300)
    np.testing.assert_array_equal(acq.LCB()(mu, std, alpha="inf"), std)


def test_gradient_x_of_kernel_trees_against_central_differences(bask):
    """``kernels.gradient_x`` -- skopt's ``kernel_.gradient_x`` (the reference's prediction gradients, ``bask/bayesgpr.py:622-635`` ->
    skopt predict) restated for scikit-learn kernel objects: every leaf and combinator against central differences of the kernel
    object itself, and the r = 0 limits."""
    from sklearn.gaussian_process.kernels import (RBF, ConstantKernel as C, DotProduct, Exponentiation, ExpSineSquared, Matern,
                                                  RationalQuadratic, WhiteKernel)

    gradient_x = bask.kernels.gradient_x
    rng = np.random.RandomState(0)
    X, x = rng.uniform(size=(9, 3)), rng.uniform(size=3)
    trees = [C(2.0) * Matern([0.3, 0.5, 0.7], nu=2.5) * RBF(0.8) + WhiteKernel(0.1),
             Matern(0.4, nu=1.5) + Matern([0.2, 0.9, 0.4], nu=0.5), C(0.7) * RationalQuadratic(0.7, 1.3), Matern(0.6, nu=0.7),
             Matern(0.6, nu=1.9), C(0.5) + ExpSineSquared(1.1, 0.8) * RBF(1.0), DotProduct(0.3) + RBF(0.5),
             Exponentiation(RBF(0.7) + C(0.2), 2.0), Matern(0.5, nu=np.inf)]
    h = 1e-6
    for k in trees:
        g = gradient_x(k, x, X)
        fd = np.empty_like(g)
        for j in range(3):
            e = np.zeros(3)
            e[j] = h
            fd[:, j] = (k((x + e)[None], X)[0] - k((x - e)[None], X)[0]) / (2 * h)
        assert g.shape == (9, 3)
        np.testing.assert_allclose(g, fd, rtol=0, atol=2e-9 * max(1.0, np.abs(fd).max()))
    for k in (Matern(0.6, nu=0.7), Matern(0.6, nu=1.9), Matern(0.4, nu=0.5), Matern(0.4, nu=2.5), RBF(0.4), RationalQuadratic(0.5, 2.0)):
        assert np.all(gradient_x(k, X[2], X)[2] == 0.0)  # at a training point the gradient with respect to that point vanishes
    with pytest.raises(NotImplementedError):
        from sklearn.gaussian_process.kernels import PairwiseKernel
        gradient_x(PairwiseKernel(), x, X)


def test_bench_counts_the_generated_gram_flops_once():
    """bench.py's split of the Gram flops the trailing update's generating launches carry (look-ahead columns of the first panel
    group, its bulk update): together with block column 0 they are the (3 d + 14) n^2 / 2 of ``lml_flops`` -- nothing counted twice."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for n, d in ((2048, 16), (1024, 8), (4096, 3), (256, 2)):
        col, bulk = bench.gram_generated_flops(n, d)
        col0 = (128 * n - 128 * 128 / 2.0) * (3 * d + 14)
        assert col >= 0 and bulk >= 0
        assert col + bulk + col0 == pytest.approx(n * n / 2.0 * (3 * d + 14))
    assert bench.gram_generated_flops(2048, 16)[0] / (3 * 16 + 14) == 128 * (1920 + 1792 + 1664) - 3 * 8192
