"""Full-size parity on DENSE, ill-conditioned matrices -- tests that can fail.

The BASELINE-size goldens of test_gpu_lml.py / test_gpu_sizes.py use the SURVEY 8(d) inputs with l ~ 0.3 in d = 8 / 16 / 32:
cond(K) = 130 / 5.7 / 1.06, nearly diagonal matrices whose log-likelihood barely depends on the panel solves and trailing
updates (zeroing every off-diagonal 128-block moves config D's by 3.7e-5 relative).  The cases here
(tests/golden/dense_sizes.npz, generated from scikit-learn 1.7.2 by tests/golden/gen_golden.py::gen_dense, which ASSERTS
that the same zeroing moves the log-likelihood by more than 1e-2 relative) are n = 2048 (d = 2) and n = 4096 (d = 3) with
length scales 0.3 .. 1.0 and noise 1e-2 .. 1e-4: cond(K) 7e4 .. 4e7, median off-diagonal entry 0.13 .. 0.84 of the diagonal.
They run on every schedule of the factorisation -- the launch schedule with P = 1 / 2 / 4 / 16 block columns per panel
group, the launch-free kernel, the automatic choice, batches of 3 and of 9 (XCD-pinned tile maps) -- and cover

* the log-likelihood (sklearn/_gpr.py:579-613 reached from bask/bayesgpr.py:374) at 1e-6;
* a batch with mixed outcomes whose failing matrices fail at a pivot in block column 13 / 23 (dpotrf info from scipy);
* the posterior: alpha, mean, variance with and without noise (bask/bayesgpr.py:200-217,622-635) at 1e-6;
* the factor itself against LAPACK's on the oracle's matrix, and its backward error;
* the 79-block-column covariance factorisation behind the Thompson draws of a config-E tell (bask/bayesgpr.py:669-678);
* and the proof that they bite: tests/fault/libbgp_fault.so -- the same sources with one 16-wide k-chunk of the trailing update
  dropped from matrix row 1536 on -- fails them on both paths while the nearly diagonal config-D golden still passes.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_variance_close, load_golden, synth

pytestmark = pytest.mark.gpu
RTOL = 1e-6

# name -> (environment read at context creation, bgp_set_persist mode or None for the automatic rule)
SCHEDULES = {
    "auto": ({}, None),
    "launches": ({}, 0),
    "launch_free": ({}, 1),
    "P1": ({"BGP_PANELS": "1"}, 0),
    "P2": ({"BGP_PANELS": "2"}, 0),
    "P4": ({"BGP_PANELS": "4"}, 0),
    "P16": ({"BGP_PANELS": "16"}, 0),
}


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


def _case(tag):
    g = load_golden("dense_sizes.npz")
    n, d, seed = [int(v) for v in g[tag + "_nd_seed"]]
    X, y = synth(n, d, seed)
    return g, n, d, X, y


def _context(lib, monkeypatch, schedule, X, y, ad, max_batch):
    env, mode = SCHEDULES[schedule]
    for k in ("BGP_PANELS", "BGP_PERSIST"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = lib.Context(X, y, ad, max_batch=max_batch)
    if mode is not None:
        ctx.set_persist(mode)
    return ctx


@pytest.mark.parametrize("schedule", list(SCHEDULES))
@pytest.mark.parametrize("tag", ["N2", "N4"])
def test_dense_lml_on_every_schedule(lib, monkeypatch, tag, schedule):
    g, n, d, X, y = _case(tag)
    assert np.all(g[tag + "_sensitivity"] > 1e-2) and np.all(g[tag + "_cond"] > 5e4)  # (what makes this test able to fail)
    TH = g[tag + "_theta"]
    ctx = _context(lib, monkeypatch, schedule, X, y, 1e-10, 9)
    got, st = ctx.lml(TH, return_status=True)
    assert np.all(st == 0)
    np.testing.assert_allclose(got, g[tag + "_lml"], rtol=RTOL)
    if schedule in ("auto", "launches", "launch_free", "P4"):
        # nine matrices: from eight on a matrix's tiles are pinned to one XCD (bgp_map_block), the launch-free kernel runs nine
        # chains; the same three vectors three times over must give the same bits three times over
        got9, st9 = ctx.lml(np.tile(TH, (3, 1)), return_status=True)
        assert np.all(st9 == 0)
        np.testing.assert_allclose(got9[:3], g[tag + "_lml"], rtol=RTOL)
        np.testing.assert_array_equal(got9[3:6], got9[:3])
        np.testing.assert_array_equal(got9[6:], got9[:3])
        np.testing.assert_array_equal(got9[:3], got)  # batch-split invariant
    if schedule == "launch_free":
        s = ctx.persist_stats()
        assert s["calls"] >= 2 and s["timeouts"] == 0 and not s["disabled"]
    ctx.close()


@pytest.mark.parametrize("schedule", ["auto", "launches", "launch_free", "P2", "P16"])
@pytest.mark.parametrize("tag", ["N2", "N4"])
def test_dense_batch_with_failures_beyond_block_column_12(lib, monkeypatch, tag, schedule):
    """One matrix of the batch factorises, two are not positive definite at a pivot in block column 13 (n = 2048) / 23
    (n = 4096): -inf and LAPACK's info for those (sklearn/_gpr.py:586-589), the right number for the survivor."""
    g, n, d, X, y = _case(tag)
    bad, val = g[tag + "_bad_index_value"]
    bad = int(bad)
    X2, ad = X.copy(), np.full(n, 1e-10)
    X2[bad] = X2[5]
    ad[bad] = val
    ctx = _context(lib, monkeypatch, schedule, X2, y, ad, 4)
    got, st = ctx.lml(g[tag + "_theta"], return_status=True)
    assert bad // 128 >= 12
    assert list(st) == [int(v) for v in g[tag + "_bad_info"]] == [0, bad + 1, bad + 1]
    assert got[1] == -np.inf and got[2] == -np.inf
    np.testing.assert_allclose(got[0], g[tag + "_bad_lml"][0], rtol=RTOL)
    ctx.close()


@pytest.mark.parametrize("tag", ["N2", "N4"])
def test_dense_posterior_mean_and_variance(lib, tag):
    """alpha = K^-1 y, predictive mean, variance with the noise and inside noise_set_to_zero against scikit-learn + the skopt
    formula with the explicit inverse.  Tolerances: 1e-6 relative (north star) -- alpha and the mean relative to their largest
    entry, the variance relative to the prior variance it is the remainder of; on top of that the variance is held to
    200 x the difference between the reference's own two ways of computing it (explicit inverse vs triangular solves,
    1.2e-10 / 1.3e-10 here: what cond(K) = 1.6e6 / 2.7e6 leaves of it on the CPU)."""
    g, n, d, X, y = _case(tag)
    th = g[tag + "_theta"][1]
    m, qseed = [int(v) for v in g[tag + "_m_qseed"]]
    Xq = np.random.RandomState(qseed).uniform(size=(m, d))
    ctx = lib.Context(X, y, 1e-10, max_batch=2)
    res = ctx.posterior(th, want_alpha=True)
    assert res["status"][0] == 0
    np.testing.assert_allclose(res["lml"][0], g[tag + "_lml"][1], rtol=RTOL)
    a = res["alpha"][0]
    ascale = np.abs(a).max()
    np.testing.assert_allclose(a[:16], g[tag + "_alpha_head"], rtol=RTOL, atol=RTOL * ascale)
    np.testing.assert_allclose(a[-16:], g[tag + "_alpha_tail"], rtol=RTOL, atol=RTOL * ascale)
    mean, var = ctx.predict(th, Xq)
    np.testing.assert_allclose(mean[0], g[tag + "_mean"], rtol=RTOL, atol=RTOL * np.abs(g[tag + "_mean"]).max())
    # the variance itself: 1e-6 relative, or -- where it is the remainder of a cancellation at cond(K) up to 4e7 -- 8 x the
    # difference between the reference's own two formulas (stored by the generator: 1.2e-10 / 1.3e-10), whichever is larger.
    # (Both reference formulas share ONE LAPACK factor, so their difference understates what a second factorisation may differ
    # by: measured 4.6 x at n = 4096; round 4 allowed 200 x.)
    assert_variance_close(var[0], g[tag + "_std"] ** 2, g[tag + "_var_selfdiff"], factor=8.0)
    th0 = th.copy()
    th0[-1] = -np.inf
    mean0, var0 = ctx.predict(th0, Xq)
    np.testing.assert_array_equal(mean0, mean)
    assert_variance_close(var0[0], g[tag + "_std0"] ** 2, g[tag + "_var_selfdiff"], factor=8.0)
    ctx.close()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("tag", ["N2", "N4"])
def test_dense_factor_against_lapack(lib, O, tag, mode):
    """The factor the LML call leaves in the workspace (bgp_debug_workspace), launch schedule and launch-free: backward error
    |L L^T - K|_max <= 64 eps |K|_max against the ORACLE's matrix, entrywise agreement with LAPACK's factor of it within
    cond(K) eps, and z = L^-1 y."""
    from scipy.linalg import cholesky, solve_triangular

    g, n, d, X, y = _case(tag)
    th = g[tag + "_theta"][1]
    ctx = lib.Context(X, y, 1e-10, max_batch=1)
    ctx.set_persist(mode)
    ctx.lml(th[None, :])
    Lw, z = ctx.debug_workspace(0)
    ctx.close()
    L = np.tril(Lw[:n, :n])
    K = O.gram_with_jitter(X, np.full(n, 1e-10), th)
    eps = np.finfo(float).eps
    assert np.abs(L @ L.T - K).max() <= 64 * eps * np.abs(K).max()
    Lref = cholesky(K, lower=True)
    cond = float(g[tag + "_cond"][1])
    assert np.abs(L - Lref).max() <= 4 * cond * eps * np.abs(Lref).max()
    zref = solve_triangular(Lref, y, lower=True)
    np.testing.assert_allclose(z[:n], zref, rtol=0, atol=4 * cond * eps * np.abs(zref).max())


def test_thompson_covariance_factor_of_79_block_columns(lib, O):
    """The covariance factorisation behind the Thompson draws of a config-E tell (bask/bayesgpr.py:669-678 -> sklearn
    sample_y): m = 10 000 candidates -> ONE 10 112 x 10 112 matrix of 79 block columns, on a covariance whose off-diagonal
    entries are not small (l = 2 in d = 8, unit noise: median |off-diagonal| 0.05 of the diagonal).  |L L^T - Sigma|_max <= 1e-12 |Sigma|_max for the factor sample_y leaves behind (bgp_debug_cov_factor; launch-free
    and by launches: the same bits) against the covariance the device builds for predict(return_cov=True) -- other kernels,
    all tiles --, that covariance against the ORACLE's at 1e-9 (cond(K) eps), and the draws are mean + L z."""
    n, d, m = 1000, 8, 10_000
    X, y = synth(n, d, 0)
    h = np.concatenate([[0.0], np.full(d, np.log(2.0)), [0.0]])
    Xq = np.random.RandomState(11).uniform(size=(m, d))
    z = np.random.RandomState(12).randn(3, m)
    hk = h.copy()
    hk[-1] = -np.inf  # the reference draws with the noise switched off (bask/bayesgpr.py:669)
    jitter = 1e-8
    outs, facs = [], []
    for mode in (1, 0):
        ctx = lib.Context(X, y, 1e-10, max_batch=2)
        ctx.set_persist(mode)
        ctx.posterior(h[None, :])
        outs.append(ctx.sample_y(0, hk, Xq, z, jitter=jitter))
        facs.append(np.tril(ctx.debug_cov_factor()[:m, :m]))
        if mode == 1:
            assert ctx.persist_stats() == {"calls": 1, "timeouts": 0, "disabled": False, "cooldown_left": 0}
        else:
            mean_dev, _, Sigma = ctx.predict(hk[None, :], Xq, return_cov=True)
            mean_dev, Sigma = mean_dev[0], Sigma[0]
        ctx.close()
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(facs[0], facs[1])
    L = facs[0]
    del facs
    smax = np.abs(Sigma).max()
    off = np.abs(Sigma[np.triu_indices(m, 1)][::997])
    assert np.median(off) > 0.02 * np.median(np.diag(Sigma))  # dense: every block column matters
    Lk, K_inv, alpha = O.posterior(X, y, np.full(n, 1e-10), h)
    Ks = O.kernel_matrix(Xq, hk, Y=X)
    So = O.kernel_matrix(Xq, hk, noise_on_diag=False)
    So -= Ks @ K_inv @ Ks.T
    assert np.abs(So - Sigma).max() <= 1e-9 * smax
    del So
    R = L @ L.T
    R -= Sigma
    R[np.diag_indices_from(R)] -= jitter
    assert np.abs(R).max() <= 1e-12 * smax
    np.testing.assert_allclose(mean_dev, Ks @ alpha, rtol=0, atol=1e-8 * max(1.0, np.abs(alpha).max()))
    np.testing.assert_allclose(outs[0], mean_dev[None, :] + z @ L.T, rtol=0, atol=1e-9 * max(1.0, np.abs(mean_dev).max()))


_FAULT_CHILD = r"""
import sys, json
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
from conftest import load_golden, synth
_lib.LIB_PATH = %(lib)r
out = {}
g = load_golden("dense_sizes.npz")
for tag in ("N2", "N4"):
    n, d, seed = [int(v) for v in g[tag + "_nd_seed"]]
    X, y = synth(n, d, seed)
    for mode in (0, 1):
        ctx = _lib.Context(X, y, 1e-10, max_batch=4)
        ctx.set_persist(mode)
        got = ctx.lml(g[tag + "_theta"])
        out["%%s_%%d" %% (tag, mode)] = (np.abs(got - g[tag + "_lml"]) / np.abs(g[tag + "_lml"])).tolist()
        ctx.close()
g2 = load_golden("lml_sizes.npz")
n, d, seed = [int(v) for v in g2["D_nd_seed"]]
X, y = synth(n, d, seed)
for mode in (0, 1):
    ctx = _lib.Context(X, y, 1e-10, max_batch=4)
    ctx.set_persist(mode)
    got = ctx.lml(g2["D_theta"])
    out["D_%%d" %% mode] = (np.abs(got - g2["D_lml"]) / np.abs(g2["D_lml"])).tolist()
    ctx.close()
print("RESULT " + json.dumps(out))
"""


def test_a_broken_trailing_update_turns_the_dense_cases_red():
    """tests/fault/libbgp_fault.so = the library's own sources with ONE 16-wide k-chunk of the trailing update dropped on the
    tiles from matrix row 1536 on (bgp_syrk4.hip, BGP_FAULT_INJECT; launch schedule and launch-free tile tasks).  Loaded in a
    child process in place of libbgp.so: every dense case is off by far more than 1e-6 on both paths -- while the nearly
    diagonal config-D golden (cond(K) = 1.06) still passes at 1e-6, which is why these cases exist."""
    fault = os.path.join(ROOT, "tests", "fault", "libbgp_fault.so")
    assert os.path.exists(fault), "tests/fault/libbgp_fault.so missing: run __graft_entry__.build() (make -C bayes-skopt_amd/csrc fault)"
    res = subprocess.run([sys.executable, "-c", _FAULT_CHILD % {"root": ROOT, "lib": fault}], capture_output=True, text=True,
                         timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for tag in ("N2", "N4"):
        for mode in (0, 1):
            err = np.array(out["%s_%d" % (tag, mode)])
            assert np.all(err > 100 * RTOL), (tag, mode, err)  # red, with two orders of magnitude to spare
    for mode in (0, 1):
        assert np.all(np.array(out["D_%d" % mode]) < RTOL), out  # the old full-size golden does not notice
