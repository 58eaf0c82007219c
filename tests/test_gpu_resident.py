"""The ensemble sampler with its state resident on the device (``bgp_mcmc_run``, include/bgp.h) against the host-driven
loop of ``sampler.EnsembleSampler.run_mcmc`` (the restatement of emcee 3.1.6 that ``bask/bayesgpr.py:510-530`` runs):
same generator stream, same moves, same chain -- the walkers' positions bit for bit, the log-probabilities to 1e-12
relative (the priors' exp / pow are the device's instead of numpy's) -- through the fused n <= 128 kernel, the launch
schedule and the launch-free factorisation; the cases that must stay on the host do."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(n, d, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(n)
    return X, y


def _sample(n, d, walkers, steps, resident, kernel=None, seed=11, **kw):
    import bayes_skopt_amd as bask
    from sklearn.gaussian_process.kernels import WhiteKernel

    X, y = _data(n, d, 4)
    if kernel is None:
        kernel = bask.construct_default_kernel(list(range(d))) + WhiteKernel(1e-2)
    gp = bask.BayesGPR(kernel=kernel, random_state=seed, normalize_y=True, resident_sampler=resident)
    gp.kernel_ = kernel.clone_with_theta(kernel.theta)
    gp.noise_ = 1e-2
    gp.sample(X, y, n_desired_samples=walkers * steps, n_burnin=0, n_walkers_per_thread=walkers, **kw)
    s = gp._sampler
    return gp, s


# (the last case: 80 proposals per half-step = two walker groups on their own streams behind the step kernel)
@pytest.mark.parametrize("n,d,walkers,steps", [(128, 2, 100, 30), (300, 3, 40, 12), (1024, 8, 32, 6), (256, 2, 160, 5)])
def test_resident_run_replays_the_host_driven_chain(n, d, walkers, steps):
    g0, s0 = _sample(n, d, walkers, steps, resident=False)
    g1, s1 = _sample(n, d, walkers, steps, resident=True)
    assert getattr(s0, "resident_runs", 0) == 0 and s1.resident_runs == 1
    c0, c1 = s0.get_chain(), s1.get_chain()
    assert c0.shape == (steps, walkers, d + 2)
    assert np.array_equal(c0, c1), "max position difference %.3e" % np.abs(c0 - c1).max()
    np.testing.assert_allclose(s1.get_log_prob(), s0.get_log_prob(), rtol=1e-12, atol=0)
    assert np.array_equal(s0.naccepted, s1.naccepted) and s0.iteration == s1.iteration
    assert s0.n_log_prob_evals == s1.n_log_prob_evals
    assert np.array_equal(g0.pos_, g1.pos_) and np.array_equal(g0.theta, g1.theta)
    # the generator ends in the same state: the NEXT run proposes the same moves in both
    a, b = s0._random.get_state(), s1._random.get_state()
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def test_resident_run_with_fixed_and_isotropic_kernel_parts():
    """The canonical map is an index table on the device: a fixed signal variance and ONE length scale for all columns."""
    from sklearn.gaussian_process.kernels import ConstantKernel, Matern, WhiteKernel

    kernel = ConstantKernel(1.5, "fixed") * Matern(length_scale=0.4, nu=2.5) + WhiteKernel(1e-2)
    _, s0 = _sample(260, 3, 20, 10, resident=False, kernel=kernel)
    _, s1 = _sample(260, 3, 20, 10, resident=True, kernel=kernel)
    assert s1.resident_runs == 1 and s1.get_chain().shape == (10, 20, 2)
    assert np.array_equal(s0.get_chain(), s1.get_chain())
    np.testing.assert_allclose(s1.get_log_prob(), s0.get_log_prob(), rtol=1e-12, atol=0)


def test_cases_that_stay_on_the_host():
    import bayes_skopt_amd as bask
    import scipy.stats as st

    # a prior the device does not know; an odd ensemble; a progress bar; warped inputs
    d = 2
    kernel = bask.construct_default_kernel([0, 1])
    from sklearn.gaussian_process.kernels import WhiteKernel

    k = kernel + WhiteKernel(1e-2)
    custom = [st.norm(0, 2).logpdf] * 4
    _, s = _sample(128, d, 20, 4, resident=True, priors=custom)
    assert getattr(s, "resident_runs", 0) == 0
    _, s = _sample(128, d, 21, 4, resident=True)
    assert getattr(s, "resident_runs", 0) == 0
    _, s = _sample(128, d, 20, 4, resident=True, progress=True)
    assert getattr(s, "resident_runs", 0) == 0
    X, y = _data(128, d, 4)
    gp = bask.BayesGPR(kernel=k, random_state=1, normalize_y=True, warp_inputs=True)
    gp.fit(X, y, n_desired_samples=40, n_burnin=1, n_walkers_per_thread=20)
    assert getattr(gp._sampler, "resident_runs", 0) == 0


def test_resident_run_sees_a_failed_factorisation_as_minus_infinity():
    """Walkers sent to a corner where K is not positive definite: the device maps status != 0 to -inf like the host path,
    the move is rejected, both chains agree."""
    import bayes_skopt_amd as bask
    from sklearn.gaussian_process.kernels import WhiteKernel

    X, y = _data(200, 2, 4)
    X[1] = X[0]  # duplicated row: singular without noise
    kernel = bask.construct_default_kernel([0, 1]) + WhiteKernel(1e-2)
    out = []
    for resident in (False, True):
        gp = bask.BayesGPR(kernel=kernel, alpha=0.0, random_state=3, normalize_y=True, resident_sampler=resident)
        gp.kernel_ = kernel.clone_with_theta(kernel.theta)
        gp.noise_ = 1e-2
        pos = np.tile(gp.kernel_.theta, (20, 1)) + 1e-2 * np.random.RandomState(0).randn(20, 4)
        pos[::2, -1] = -80.0  # noise variance e^-80: K is numerically singular for every second walker
        gp.sample(X, y, n_desired_samples=20 * 8, n_burnin=0, n_walkers_per_thread=20, position=pos)
        out.append((gp._sampler.get_chain(), gp._sampler.get_log_prob()))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(np.isneginf(out[0][1]), np.isneginf(out[1][1])) and np.isneginf(out[0][1]).any()


_CHILD = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np
import bayes_skopt_amd as bask
from sklearn.gaussian_process.kernels import WhiteKernel
rng = np.random.RandomState(4); X = rng.uniform(size=(1024, 4)); y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(1024)
kernel = bask.construct_default_kernel([0, 1, 2, 3]) + WhiteKernel(1e-2)
gp = bask.BayesGPR(kernel=kernel, random_state=5, normalize_y=True, resident_sampler=(sys.argv[1] == "1"))
gp.kernel_ = kernel.clone_with_theta(kernel.theta); gp.noise_ = 1e-2
gp.sample(X, y, n_desired_samples=16 * 5, n_burnin=0, n_walkers_per_thread=16)
s = gp._sampler
print("RESULT " + json.dumps({"chain": [float(v).hex() for v in s.get_chain().ravel()], "resident": getattr(s, "resident_runs", 0),
                              "stats": gp._ctx.persist_stats()}))
"""


def test_a_launch_free_time_out_inside_the_resident_run_redoes_it_on_the_launch_schedule():
    def run(resident, env):
        res = subprocess.run([sys.executable, "-c", _CHILD % ROOT, "1" if resident else "0"], env=dict(os.environ, **env),
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:]), res.stderr

    ref, _ = run(False, {"BGP_PERSIST": "0"})
    got, err = run(True, {"BGP_PERSIST": "1"})
    assert got["resident"] == 1 and got["chain"] == ref["chain"] and "timed out" not in err
    assert got["stats"]["calls"] >= 10 and got["stats"]["timeouts"] == 0
    got, err = run(True, {"BGP_PERSIST": "1", "BGP_PS_TIMEOUT_TICKS": "200"})
    assert got["resident"] == 1 and got["chain"] == ref["chain"]
    assert err.count("timed out") == 1 and got["stats"]["timeouts"] == 1, err[-1500:]
