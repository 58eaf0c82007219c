#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ -- runs ONLY in the build container.

Sources of truth (nothing here travels to the GPU box except the .npz outputs):
  * scikit-learn 1.7.2 (``uv.lock:2411`` pin of the reference; it is the arithmetic
    ``bask/bayesgpr.py:374`` reaches): GaussianProcessRegressor.log_marginal_likelihood /
    kernels / predict.
  * the reference's own modules imported from /root/reference with the tier-1 shim of
    SURVEY.md Appendix A (P2): bask.priors, bask.utils, bask.acquisition.  ``bask/__init__``
    is never executed (emcee / skopt / arviz are absent from this image).

Usage:  python tests/golden/gen_golden.py      (writes tests/golden/*.npz)
"""
import importlib
import os
import sys
import types

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular
from sklearn.gaussian_process import GaussianProcessRegressor
from sklearn.gaussian_process import kernels as sk

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# ---------------------------------------------------------------------------------------
def import_reference_tier1():
    """SURVEY.md Appendix A, P2."""
    for name in ("skopt", "skopt.learning", "skopt.learning.gaussian_process"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skopt.learning.gaussian_process.kernels"] = sk
    pkg = types.ModuleType("bask")
    pkg.__path__ = [os.path.join(REF, "bask")]
    sys.modules["bask"] = pkg
    mods = {}
    for m in ("priors", "init", "utils", "acquisition"):
        mods[m] = importlib.import_module("bask." + m)
    return mods


def synth(n, d, seed):
    """SURVEY.md 8(d) synthetic inputs."""
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    return X, y


def make_kernel(stationary, form, d, ard=True):
    ls = [0.3] * d if ard else 0.3
    if stationary == "rbf":
        S = sk.RBF(length_scale=ls)
    else:
        nu = {"matern12": 0.5, "matern32": 1.5, "matern52": 2.5}[stationary]
        S = sk.Matern(length_scale=ls, nu=nu)
    C = sk.ConstantKernel(1.0)
    k = C * S if form == "product" else C + S
    return k + sk.WhiteKernel()


def thetas(d, B, seed, spread=0.3):
    rng = np.random.RandomState(seed)
    base = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
    return base + spread * rng.randn(B, d + 2)


def sk_gpr(kernel, X, y, alpha):
    g = GaussianProcessRegressor(kernel=kernel, optimizer=None, alpha=alpha)
    g.fit(X, y)
    return g


# ---------------------------------------------------------------------------------------
def gen_lml_small(out):
    """Every kernel form at small n: K, LML, alpha, diag(L)."""
    rec = {}
    case = 0
    for n, d in ((24, 2), (40, 3)):
        X, y = synth(n, d, seed=100 + n)
        avec = 1e-10 + 0.05 * np.random.RandomState(7).rand(n)
        for stationary in ("rbf", "matern12", "matern32", "matern52"):
            for form in ("product", "sum"):
                for alpha in (1e-10, avec):
                    k = make_kernel(stationary, form, d)
                    g = sk_gpr(k, X, y, alpha)
                    TH = thetas(d, 6, seed=case)
                    Ks, lmls, alphas, Ldiag = [], [], [], []
                    for th in TH:
                        kk = g.kernel_.clone_with_theta(th)
                        K = kk(X)
                        Kj = K.copy()
                        Kj[np.diag_indices_from(Kj)] += alpha
                        L = cholesky(Kj, lower=True)
                        Ks.append(Kj)
                        Ldiag.append(np.diag(L))
                        alphas.append(cho_solve((L, True), y))
                        lmls.append(g.log_marginal_likelihood(th))
                    pre = f"c{case}_"
                    rec[pre + "X"] = X
                    rec[pre + "y"] = y
                    rec[pre + "alpha_diag"] = np.broadcast_to(alpha, (n,)).copy()
                    rec[pre + "theta"] = TH
                    rec[pre + "K"] = np.array(Ks)
                    rec[pre + "lml"] = np.array(lmls)
                    rec[pre + "alpha_vec"] = np.array(alphas)
                    rec[pre + "Ldiag"] = np.array(Ldiag)
                    rec[pre + "meta"] = np.array([stationary, form])
                    case += 1
    rec["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(out, "lml_small.npz"), **rec)
    print("lml_small:", case, "cases")


def gen_lml_sizes(out):
    """Config A at full size with data; mid-size ragged n; B/C/D as seed + scalars."""
    rec = {}
    # A: n=128, d=2
    X, y = synth(128, 2, seed=0)
    g = sk_gpr(make_kernel("matern52", "product", 2), X, y, 1e-10)
    TH = thetas(2, 16, seed=11)
    # edge cases: tiny / huge length scale, tiny noise
    TH[12, 1:3] = np.log(0.01)
    TH[13, 1:3] = np.log(10.0)
    TH[14, 3] = np.log(1e-10)
    TH[15, 0] = np.log(50.0)
    rec["A_X"], rec["A_y"], rec["A_theta"] = X, y, TH
    rec["A_lml"] = np.array([g.log_marginal_likelihood(t) for t in TH])
    # ragged mid size with vector alpha: n=300,d=5 and n=777,d=3
    for tag, n, d, seed in (("M1", 300, 5, 1), ("M2", 777, 3, 2), ("M3", 129, 4, 3), ("M4", 257, 1, 4)):
        X, y = synth(n, d, seed)
        avec = 1e-10 + 0.02 * np.random.RandomState(seed).rand(n)
        g = sk_gpr(make_kernel("matern52", "product", d), X, y, avec)
        TH = thetas(d, 5, seed=20 + seed)
        rec[tag + "_nd_seed"] = np.array([n, d, seed])
        rec[tag + "_alpha_diag"] = avec
        rec[tag + "_theta"] = TH
        rec[tag + "_lml"] = np.array([g.log_marginal_likelihood(t) for t in TH])
    # B / C / D: seeds + scalars only (X regenerated by synth())
    for tag, n, d, nb in (("B", 1024, 8, 4), ("C", 2048, 16, 3), ("D", 4096, 32, 2)):
        X, y = synth(n, d, seed=0)
        g = sk_gpr(make_kernel("matern52", "product", d), X, y, 1e-10)
        TH = thetas(d, nb, seed=30, spread=0.2)
        rec[tag + "_nd_seed"] = np.array([n, d, 0])
        rec[tag + "_theta"] = TH
        rec[tag + "_lml"] = np.array([g.log_marginal_likelihood(t) for t in TH])
        print(tag, rec[tag + "_lml"])
    # exactly singular: duplicated rows, alpha = 0, no noise -> -inf (sklearn:_gpr.py:588-589)
    X, y = synth(16, 2, seed=5)
    X[1] = X[0]
    k = sk.ConstantKernel(1.0) * sk.Matern([0.3, 0.3], nu=2.5) + sk.WhiteKernel(1.0)
    g = sk_gpr(k, X, y, 0.0)
    th = np.array([0.0, np.log(0.3), np.log(0.3), -np.inf])
    with np.errstate(all="ignore"):
        val = g.log_marginal_likelihood(np.array([0.0, np.log(0.3), np.log(0.3), -745.0]))
    rec["S_X"], rec["S_y"], rec["S_theta"], rec["S_lml"] = X, y, th[None], np.array([val])
    assert val == -np.inf, val
    np.savez_compressed(os.path.join(out, "lml_sizes.npz"), **rec)


def gen_grad(out):
    rec = {}
    case = 0
    for stationary in ("rbf", "matern32", "matern52"):
        for form in ("product", "sum"):
            n, d = 50, 3
            X, y = synth(n, d, seed=40 + case)
            g = sk_gpr(make_kernel(stationary, form, d), X, y, 1e-10)
            TH = thetas(d, 3, seed=50 + case)
            vals, grads = [], []
            for t in TH:
                v, gr = g.log_marginal_likelihood(t, eval_gradient=True)
                vals.append(v)
                grads.append(gr)
            pre = f"c{case}_"
            rec[pre + "X"], rec[pre + "y"], rec[pre + "theta"] = X, y, TH
            rec[pre + "lml"], rec[pre + "grad"] = np.array(vals), np.array(grads)
            rec[pre + "meta"] = np.array([stationary, form])
            case += 1
    rec["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(out, "lml_grad.npz"), **rec)


def skopt_predict(g, K_inv, X, return_cov=False):
    """skopt 0.10.2 GaussianProcessRegressor.predict restated on a fitted
    sklearn GPR: mean = K_trans alpha_; var = diag - einsum(K_trans, K_trans, K_inv_)."""
    K_trans = g.kernel_(X, g.X_train_)
    mean = K_trans.dot(g.alpha_)
    var = g.kernel_.diag(X).copy()
    var -= np.einsum("ki,kj,ij->k", K_trans, K_trans, K_inv)
    var[var < 0] = 0.0
    out = [mean, np.sqrt(var)]
    if return_cov:
        v = cho_solve((g.L_, True), K_trans.T)
        out.append(g.kernel_(X) - K_trans.dot(v))
    return out


def gen_predict(out):
    rec = {}
    case = 0
    for stationary, form, n, d, m in (("matern52", "product", 60, 3, 40), ("rbf", "product", 33, 1, 17),
                                      ("matern32", "sum", 45, 2, 25), ("matern52", "product", 200, 4, 64)):
        X, y = synth(n, d, seed=60 + case)
        Xq = np.random.RandomState(70 + case).uniform(size=(m, d))
        avec = 1e-10 + 0.03 * np.random.RandomState(case).rand(n)
        for alpha in (1e-10, avec):
            th = thetas(d, 1, seed=80 + case, spread=0.2)[0]
            k = make_kernel(stationary, form, d).clone_with_theta(th)
            g = sk_gpr(k, X, y, alpha)
            L_inv = solve_triangular(g.L_.T, np.eye(n))
            K_inv = L_inv.dot(L_inv.T)
            mean, std, cov = skopt_predict(g, K_inv, Xq, return_cov=True)
            # cross-check against sklearn's own predict
            m2, s2 = g.predict(Xq, return_std=True)
            assert np.allclose(mean, m2, rtol=1e-9, atol=1e-12)
            assert np.allclose(std, s2, rtol=1e-5, atol=1e-7), np.abs(std - s2).max()
            # noise_set_to_zero: swap the white kernel for WhiteKernel(0) WITHOUT refitting
            g.kernel_.set_params(k2=sk.WhiteKernel(noise_level=0.0))
            mean0, std0, cov0 = skopt_predict(g, K_inv, Xq, return_cov=True)
            # the reference's own uncertainty about a predictive variance: skopt's formula (einsum with the explicit inverse,
            # above) against scikit-learn's (triangular solve, sklearn/_gpr.py:470-489) on the same factor, with and without noise
            K_trans = g.kernel_(Xq, g.X_train_)
            V = solve_triangular(g.L_, K_trans.T, lower=True)
            var0_sk = g.kernel_.diag(Xq) - np.einsum("ij,ij->j", V, V)
            pre = f"c{case}_"
            rec[pre + "var0_selfdiff"] = np.array(np.abs(std0 ** 2 - np.maximum(var0_sk, 0.0)).max())
            rec[pre + "var_selfdiff"] = np.array(np.abs(std ** 2 - s2 ** 2).max())
            rec[pre + "X"], rec[pre + "y"], rec[pre + "Xq"] = X, y, Xq
            rec[pre + "alpha_diag"] = np.broadcast_to(alpha, (n,)).copy()
            rec[pre + "theta"] = th
            rec[pre + "mean"], rec[pre + "std"], rec[pre + "cov"] = mean, std, cov
            rec[pre + "mean0"], rec[pre + "std0"], rec[pre + "cov0"] = mean0, std0, cov0
            rec[pre + "L"], rec[pre + "K_inv"], rec[pre + "alpha_vec"] = g.L_, K_inv, g.alpha_
            rec[pre + "meta"] = np.array([stationary, form])
            case += 1
    rec["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(out, "predict.npz"), **rec)
    print("predict:", case, "cases")


def gen_reference_tier1(out):
    ref = import_reference_tier1()
    rec = {}
    # --- priors: known answers of tests/test_utils.py:20-40 + a grid
    kernel = sk.ConstantKernel(1.0, (0.1, 2.0)) * sk.Matern([0.3, 0.3], (0.2, 0.5), nu=2.5) + sk.WhiteKernel()
    pri = ref["utils"].guess_priors(kernel)
    assert len(pri) == 4
    grid = np.linspace(-8.0, 3.0, 111)
    rec["prior_grid"] = grid
    rec["prior_variance"] = np.array([pri[0](t) for t in grid])
    rec["prior_lengthscale"] = np.array([pri[1](t) for t in grid])
    rec["prior_noise"] = np.array([pri[3](t) for t in grid])
    rec["prior_known_x"] = np.array(-0.9)
    rec["prior_known_roundflat"] = np.array(pri[1](-0.9))
    rec["prior_known_halfnorm"] = np.array(pri[0](-0.9))
    assert abs(pri[1](-0.9) - (-0.02116327824572739)) < 1e-12
    assert abs(pri[0](-0.9) - (-2.112906921232193)) < 1e-12
    rf = ref["priors"].make_roundflat()
    rec["roundflat_x"] = np.linspace(0.02, 2.0, 100)
    rec["roundflat_val"] = np.array([rf(t) for t in rec["roundflat_x"]])
    rf2 = ref["priors"].make_roundflat(0.2, 0.9, 3.0, 4.0)
    rec["roundflat2_val"] = np.array([rf2(t) for t in rec["roundflat_x"]])
    # --- geometric median
    rng = np.random.RandomState(3)
    for i, (S, p) in enumerate(((200, 4), (1000, 10), (50, 2))):
        C = rng.randn(S, p) * rng.rand(p) + rng.randn(p)
        if i == 2:
            C[:20] = C[0]  # repeated rows -> exercises the num_zeros branch
        rec[f"gm{i}_chain"] = C
        rec[f"gm{i}_median"] = ref["utils"].geometric_median(C)
    # --- default kernel (bask/utils.py:127-151)
    dk = ref["utils"].construct_default_kernel([0, 1, 2])
    rec["default_kernel_theta"] = dk.theta
    rec["default_kernel_bounds"] = dk.bounds
    # --- closed-form acquisitions on fixed (mu, std)
    acq = ref["acquisition"]
    mu = rng.randn(64)
    std = np.abs(rng.randn(64)) * 0.5
    std[:3] = 0.0
    rec["acq_mu"], rec["acq_std"] = mu, std
    rec["acq_ei"] = acq.ExpectedImprovement()(mu, std)
    rec["acq_ei_yopt"] = acq.ExpectedImprovement()(mu, std, y_opt=-0.3)
    rec["acq_lcb"] = acq.LCB()(mu, std)
    rec["acq_lcb3"] = acq.LCB()(mu, std, alpha=3.0)
    rec["acq_mean"] = acq.Expectation()(mu, std)
    rec["acq_ttei"] = acq.TopTwoEI()(mu, std)

    # --- PVRS / VR with a duck-typed GP and injected Thompson samples
    class DuckGP:
        warp_inputs = False

        def __init__(self, X, kernel_, alpha, thompson):
            self.X_train_, self.kernel_, self.alpha, self._th = X, kernel_, alpha, thompson

        def sample_y(self, X, sample_mean=True, n_samples=1, random_state=None):
            return self._th

    case = 0
    for n, d, m, T, vec in ((30, 2, 50, 5, True), (64, 3, 40, 10, False), (130, 2, 33, 4, True)):
        X, y = synth(n, d, seed=90 + case)
        Xc = np.random.RandomState(95 + case).uniform(size=(m, d))
        th = thetas(d, 1, seed=97 + case, spread=0.2)[0]
        k = make_kernel("matern52", "product", d).clone_with_theta(th)
        alpha = (1e-10 + 0.01 * np.random.RandomState(case).rand(n)) if vec else 1e-10
        thompson = np.random.RandomState(99 + case).randn(m, T)
        gp = DuckGP(X, k, alpha, thompson)
        covs = acq.PVRS()(Xc, gp, n_thompson=T, random_state=0)
        pre = f"pvrs{case}_"
        rec[pre + "X"], rec[pre + "Xc"], rec[pre + "theta"] = X, Xc, th
        rec[pre + "alpha_vec"] = alpha if vec else np.array([])
        rec[pre + "thompson"] = thompson
        rec[pre + "covs"] = covs
        if case < 2:
            rec[pre + "vr"] = acq.VarianceReduction()(Xc, gp)
        case += 1
    rec["pvrs_cases"] = np.array(case)
    np.savez_compressed(os.path.join(out, "reference_tier1.npz"), **rec)
    print("tier1 ok")


def gen_sizes_posterior(out):
    """Posterior mean / std at the BASELINE sizes (configs B, C, D: bask/bayesgpr.py:622-635 -> skopt predict,
    SURVEY.md 3.4) and the reference's own PVRS on a candidate subset at config E's size
    (bask/acquisition.py:316-339).  Seeds + scalars only: X, y, Xq are regenerated by synth() / RandomState."""
    ref = import_reference_tier1()
    rec = {}
    for tag, n, d in (("B", 1024, 8), ("C", 2048, 16), ("D", 4096, 32)):
        X, y = synth(n, d, seed=0)
        m = 64
        Xq = np.random.RandomState(200 + n).uniform(size=(m, d))
        th = thetas(d, 1, seed=210 + n, spread=0.15)[0]
        g = sk_gpr(make_kernel("matern52", "product", d).clone_with_theta(th), X, y, 1e-10)
        L_inv = solve_triangular(g.L_.T, np.eye(n))
        K_inv = L_inv.dot(L_inv.T)  # the explicit inverse the reference forms (bask/bayesgpr.py:214-216)
        mean, std = skopt_predict(g, K_inv, Xq)
        m2, s2 = g.predict(Xq, return_std=True)
        assert np.allclose(mean, m2, rtol=1e-8, atol=1e-11)
        g.kernel_.set_params(k2=sk.WhiteKernel(noise_level=0.0))  # noise_set_to_zero, no refit
        mean0, std0 = skopt_predict(g, K_inv, Xq)
        rec[tag + "_nd_seed_m_qseed"] = np.array([n, d, 0, m, 200 + n])
        rec[tag + "_theta"] = th
        rec[tag + "_mean"], rec[tag + "_std"], rec[tag + "_std0"] = mean, std, std0
        rec[tag + "_lml"] = np.array(sk_gpr(make_kernel("matern52", "product", d), X, y, 1e-10).log_marginal_likelihood(th))
        rec[tag + "_alpha_head"] = g.alpha_[:16]
        print(tag, "mean[:3]", mean[:3], "std[:3]", std[:3], "cond(K) ~ %.2e" % np.linalg.cond(g.L_) ** 2)

    # config E size: n = 1024, d = 8, vector alpha (Optimizer.tell always passes one), 64 candidates, 8 Thompson draws
    class DuckGP:
        warp_inputs = False

        def __init__(self, X, kernel_, alpha, thompson):
            self.X_train_, self.kernel_, self.alpha, self._th = X, kernel_, alpha, thompson

        def sample_y(self, X, sample_mean=True, n_samples=1, random_state=None):
            return self._th

    n, d, m, T = 1024, 8, 64, 8
    X, y = synth(n, d, seed=0)
    Xc = np.random.RandomState(300).uniform(size=(m, d))
    th = thetas(d, 1, seed=301, spread=0.15)[0]
    k = make_kernel("matern52", "product", d).clone_with_theta(th)
    alpha = 1e-10 + 0.01 * np.random.RandomState(302).rand(n)
    thompson = np.random.RandomState(303).randn(m, T)
    covs = ref["acquisition"].PVRS()(Xc, DuckGP(X, k, alpha, thompson), n_thompson=T, random_state=0)
    rec["E_nd_seed_m_T"] = np.array([n, d, 0, m, T])
    rec["E_theta"], rec["E_covs"] = th, covs
    print("E covs[:4]", covs[:4])
    np.savez_compressed(os.path.join(out, "posterior_sizes.npz"), **rec)



def zero_offdiag_blocks(K, nb=128):
    """K with every off-diagonal nb x nb block zeroed: what a blocked factorisation whose panel solves and trailing updates
    do nothing at all would see."""
    Z = np.zeros_like(K)
    for i in range(0, K.shape[0], nb):
        Z[i:i + nb, i:i + nb] = K[i:i + nb, i:i + nb]
    return Z


def lml_of(K, y):
    L = cholesky(K, lower=True)
    a = cho_solve((L, True), y)
    return -0.5 * y.dot(a) - np.log(np.diag(L)).sum() - 0.5 * len(y) * np.log(2.0 * np.pi)


def gen_dense(out):
    """DENSE, ill-conditioned matrices at the sizes where the blocked paths switch on (P = 4 panel groups from 12 block
    columns, the 16-32-block-column index ranges of the trailing update / panel solve / launch-free tile tasks): n = 2048
    (d = 2) and n = 4096 (d = 3), length scales 0.3 .. 1.0, noise 1e-4 .. 1e-2 -> cond(K) 1e4 .. 1e7, median off-diagonal
    entry O(0.1) of the diagonal.  (The B / C / D cases above use the 8(d) inputs with l ~ 0.3 in d = 8 / 16 / 32: cond(K)
    = 130 / 5.7 / 1.06 -- nearly diagonal; a factorisation with a broken trailing update passes them at 1e-6.)
    Every case ASSERTS its own sensitivity: the log-likelihood with all off-diagonal 128-blocks of K zeroed must differ by
    more than 1e-2 relative, so a nearly diagonal case cannot be committed here again.
    Stored: seeds + scalars only (X, y, Xq regenerated by synth() / RandomState): log-likelihoods of three hyper-parameter
    vectors (sklearn/_gpr.py:579-613 reached from bask/bayesgpr.py:374), alpha head, posterior mean / std / noise-free std at
    64 query points for the middle one (skopt predict formula with the explicit inverse, bask/bayesgpr.py:200-217,622-635),
    and a batch with mixed outcomes: a coinciding pair of points whose second member lies beyond block column 12 and carries
    a NEGATIVE per-point diagonal term (Optimizer.tell always passes a vector alpha): with enough noise the matrix
    factorises, otherwise LAPACK's dpotrf fails exactly at that pivot (-inf, sklearn/_gpr.py:586-589)."""
    rec = {}
    for tag, n, d, bad in (("N2", 2048, 2, 1700), ("N4", 4096, 3, 3000)):
        X, y = synth(n, d, seed=7)
        ells = (0.3, 0.6, 1.0)
        sig2 = (1e-2, 1e-3, 1e-4)
        TH = np.array([np.concatenate([[np.log(1.0 + 0.2 * i)], np.log(ells[i] * (1.0 + 0.1 * np.arange(d))), [np.log(sig2[i])]])
                       for i in range(3)])
        g = sk_gpr(make_kernel("matern52", "product", d), X, y, 1e-10)
        lmls, conds, meds, sens = [], [], [], []
        for th in TH:
            K = g.kernel_.clone_with_theta(th)(X)
            K[np.diag_indices_from(K)] += 1e-10
            lml = g.log_marginal_likelihood(th)
            assert abs(lml_of(K, y) - lml) <= 1e-9 * abs(lml)
            ev = np.linalg.eigvalsh(K)
            off = np.abs(K[np.triu_indices(n, 1)])
            lz = lml_of(zero_offdiag_blocks(K), y)
            lmls.append(lml)
            conds.append(ev[-1] / ev[0])
            meds.append(np.median(off) / K[0, 0])
            sens.append(abs(lz - lml) / abs(lml))
            print(tag, "theta", np.round(th, 3), "lml %.6f cond %.2e median offdiag/diag %.3f sensitivity %.3f"
                  % (lml, conds[-1], meds[-1], sens[-1]))
            assert sens[-1] > 1e-2, "nearly diagonal case: the blocked machinery would not be tested"
            assert meds[-1] > 0.02
        rec[tag + "_nd_seed"] = np.array([n, d, 7])
        rec[tag + "_theta"] = TH
        rec[tag + "_lml"] = np.array(lmls)
        rec[tag + "_cond"] = np.array(conds)
        rec[tag + "_sensitivity"] = np.array(sens)
        # posterior at the middle vector
        th = TH[1]
        m = 64
        Xq = np.random.RandomState(500 + n).uniform(size=(m, d))
        gp = sk_gpr(make_kernel("matern52", "product", d).clone_with_theta(th), X, y, 1e-10)
        L_inv = solve_triangular(gp.L_.T, np.eye(n))
        K_inv = L_inv.dot(L_inv.T)
        mean, std = skopt_predict(gp, K_inv, Xq)
        m2, s2 = gp.predict(Xq, return_std=True)
        assert np.allclose(mean, m2, rtol=1e-7, atol=1e-9)
        gp.kernel_.set_params(k2=sk.WhiteKernel(noise_level=0.0))
        mean0, std0 = skopt_predict(gp, K_inv, Xq)
        rec[tag + "_m_qseed"] = np.array([m, 500 + n])
        rec[tag + "_alpha_head"] = gp.alpha_[:16]
        rec[tag + "_alpha_tail"] = gp.alpha_[-16:]
        rec[tag + "_mean"], rec[tag + "_std"], rec[tag + "_std0"] = mean, std, std0
        # how well the reference's own arithmetic knows these numbers: explicit-inverse formula vs sklearn's solve
        rec[tag + "_var_selfdiff"] = np.array(np.abs(std ** 2 - s2 ** 2).max())
        print(tag, "mean[:3]", mean[:3], "std[:3]", std[:3], "std0[:3]", std0[:3], "var self-diff %.2e" % rec[tag + "_var_selfdiff"])
        # a batch with MIXED outcomes: point `bad` (block column bad // 128 >= 12) is a copy of point 5 and carries a negative
        # diagonal term -3e-3.  Its pivot lies in [s2 - 3e-3, 2 s2 - 3e-3 + 1e-10] (s2 = the vector's noise): the vector
        # with s2 = 1e-2 factorises, the two others fail exactly there (LinAlgError -> -inf, sklearn/_gpr.py:586-589)
        X2 = X.copy()
        X2[bad] = X2[5]
        avec = np.full(n, 1e-10)
        avec[bad] = -3e-3
        gb = sk_gpr(make_kernel("matern52", "product", d), X[:8], y[:8], 1e-10)  # (a fit with this alpha could raise)
        gb.X_train_, gb.y_train_, gb.alpha = X2, y, avec
        with np.errstate(all="ignore"):
            vals = np.array([gb.log_marginal_likelihood(t) for t in TH])
        assert np.isfinite(vals[0]) and vals[1] == -np.inf and vals[2] == -np.inf, vals
        infos = [0]
        for t in TH[1:]:
            K = g.kernel_.clone_with_theta(t)(X2)
            K[np.diag_indices_from(K)] += avec
            try:
                cholesky(K, lower=True)
                raise AssertionError("expected LinAlgError")
            except np.linalg.LinAlgError as e:
                infos.append(int(str(e).split("-th")[0]))
        assert infos[1] == bad + 1 and infos[2] == bad + 1 and bad // 128 >= 12, (infos, bad)
        rec[tag + "_bad_index_value"] = np.array([bad, -3e-3])
        rec[tag + "_bad_lml"] = vals
        rec[tag + "_bad_info"] = np.array(infos)
        print(tag, "mixed case:", vals, "dpotrf info", infos, "(block column %d)" % (bad // 128))
    np.savez_compressed(os.path.join(out, "dense_sizes.npz"), **rec)


def gen_sample_y(out):
    """a8: the call the reference makes for function draws -- sklearn's GaussianProcessRegressor.sample_y
    (sklearn/_gpr.py:522-526: numpy's legacy multivariate_normal, i.e. an SVD of the predictive covariance), reached from
    bask/bayesgpr.py:669-678.  Stored: the distribution's parameters as sklearn computes them (predict with return_cov)
    and the sample mean / covariance of 4000 reference draws: the device draws (Cholesky factor, other variates) must
    describe the same distribution within Monte-Carlo error."""
    n, d, m, ndraw = 60, 2, 6, 4000
    X, y = synth(n, d, 77)
    k = sk.ConstantKernel(1.3) * sk.Matern(length_scale=[0.35, 0.5], nu=2.5) + sk.WhiteKernel(0.02)
    gpr = GaussianProcessRegressor(kernel=k, optimizer=None, alpha=1e-10).fit(X, y)
    Xq = np.random.RandomState(78).uniform(size=(m, d))
    # the reference draws with the noise switched off (bask/bayesgpr.py:669: noise_set_to_zero)
    k0 = sk.ConstantKernel(1.3) * sk.Matern(length_scale=[0.35, 0.5], nu=2.5) + sk.WhiteKernel(1e-300)
    mean, cov = gpr.predict(Xq, return_cov=True)
    gpr0 = GaussianProcessRegressor(kernel=k0, optimizer=None, alpha=1e-10)
    gpr0.fit(X, y)
    gpr0.L_, gpr0.alpha_ = gpr.L_, gpr.alpha_   # factors of the noisy kernel, noise-free prior at the query points
    mean0, cov0 = gpr0.predict(Xq, return_cov=True)
    draws = gpr0.sample_y(Xq, n_samples=ndraw, random_state=5)   # (m, ndraw)
    np.savez(os.path.join(out, "sample_y.npz"), X=X, y=y, Xq=Xq, theta=gpr.kernel_.theta, mean=mean0, cov=cov0,
             ref_sample_mean=draws.mean(axis=1), ref_sample_cov=np.cov(draws), ndraw=ndraw,
             mean_noisy=mean, cov_noisy=cov)


def gen_mvn(out):
    """SURVEY 8c golden (8): the reference's OWN seeded function draws -- scikit-learn's sample_y (sklearn/_gpr.py:522-526:
    predict(return_cov=True), then numpy's legacy RandomState.multivariate_normal, i.e. standard normals times the SVD factor of
    the covariance) as bask/bayesgpr.py:669-678 reaches it: noise switched off (case "nf": WhiteKernel(0) at the query points,
    factors of the noisy kernel) or left on (case "ny").  m = 6, 24 and 64 query points; the generator state is one
    RandomState(5) per call, 3 draws each.  The smallest singular value of every covariance is stored: directions whose singular
    value is numerically zero are arbitrary in ANY implementation and enter a draw with weight sqrt(s)."""
    n, d = 60, 2
    X, y = synth(n, d, 77)
    k = sk.ConstantKernel(1.3) * sk.Matern(length_scale=[0.35, 0.5], nu=2.5) + sk.WhiteKernel(0.02)
    gpr = GaussianProcessRegressor(kernel=k, optimizer=None, alpha=1e-10).fit(X, y)
    k0 = sk.ConstantKernel(1.3) * sk.Matern(length_scale=[0.35, 0.5], nu=2.5) + sk.WhiteKernel(1e-300)
    gpr0 = GaussianProcessRegressor(kernel=k0, optimizer=None, alpha=1e-10)
    gpr0.fit(X, y)
    gpr0.L_, gpr0.alpha_ = gpr.L_, gpr.alpha_   # factors of the noisy kernel, noise-free prior at the query points
    rec = dict(X=X, y=y, theta=gpr.kernel_.theta, seed=5, ndraw=3)
    for m in (6, 24, 64):
        Xq = np.random.RandomState(100 + m).uniform(size=(m, d))
        rec["Xq_%d" % m] = Xq
        for tag, g in (("nf", gpr0), ("ny", gpr)):
            mean, cov = g.predict(Xq, return_cov=True)
            draws = g.sample_y(Xq, n_samples=3, random_state=5)      # (m, 3)
            again = np.random.RandomState(5).multivariate_normal(mean, cov, 3).T
            assert np.array_equal(draws, again)
            sv = np.linalg.svd(cov, compute_uv=False)
            rec["%s_draws_%d" % (tag, m)] = draws
            rec["%s_mean_%d" % (tag, m)] = mean
            rec["%s_cov_%d" % (tag, m)] = cov
            rec["%s_smin_%d" % (tag, m)] = sv.min()
            print("mvn", tag, m, "singular values %.3g .. %.3g" % (sv.max(), sv.min()))
    np.savez_compressed(os.path.join(out, "mvn.npz"), **rec)


if __name__ == "__main__":
    if len(sys.argv) > 1:   # e.g.  python tests/golden/gen_golden.py gen_mvn
        for name in sys.argv[1:]:
            globals()[name](HERE)
        print("done")
        sys.exit(0)
    gen_lml_small(HERE)
    gen_lml_sizes(HERE)
    gen_grad(HERE)
    gen_predict(HERE)
    gen_reference_tier1(HERE)
    gen_sizes_posterior(HERE)
    gen_sample_y(HERE)
    gen_mvn(HERE)
    gen_dense(HERE)
    print("done")
