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
# (... and BASELINE config C's own shape: n = 2048, d = 16, 256 walkers -- 128 matrices per half-step on two walker-group streams)
@pytest.mark.parametrize("n,d,walkers,steps", [(128, 2, 100, 30), (300, 3, 40, 12), (1024, 8, 32, 6), (256, 2, 160, 5), (2048, 16, 256, 2)])
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


def test_cases_that_stay_on_the_host(capfd):
    """What is left for the host-driven loop: a prior the device cannot evaluate, an odd ensemble, a kernel tree without a canonical
    device form -- and each says so on stderr, once per process.  (Progress bars, warped inputs and ensembles beyond the step kernel's LDS run resident: below.)"""
    from bayes_skopt_amd import sampler as sampler_module

    sampler_module._told.clear()  # (once per PROCESS: earlier tests of this process may have said the same things)
    capfd.readouterr()
    d = 2
    _, s = _sample(128, d, 20, 4, resident=True, priors=[lambda t: -0.5 * t * t] * 4)
    assert getattr(s, "resident_runs", 0) == 0
    _, s = _sample(128, d, 20, 4, resident=True, priors=[lambda t: -0.5 * t * t] * 4)
    assert getattr(s, "resident_runs", 0) == 0
    _, s = _sample(128, d, 21, 4, resident=True)
    assert getattr(s, "resident_runs", 0) == 0
    from sklearn.gaussian_process.kernels import Matern

    _, s = _sample(128, d, 20, 4, resident=True, kernel=Matern(length_scale=0.5, nu=2.5) + Matern(length_scale=2.0, nu=1.5),
                   priors=[lambda t: -0.5 * t * t] * 2)
    assert getattr(s, "resident_runs", 0) == 0
    err = capfd.readouterr().err
    assert err.count("driven from the host") == 3 and "not one of the families" in err and "odd number of walkers" in err
    assert "no canonical device form" in err
    # asked for: not a fallback, nothing to say
    _, s = _sample(128, d, 20, 4, resident=False)
    assert getattr(s, "resident_runs", 0) == 0 and "driven from the host" not in capfd.readouterr().err


def test_default_fit_with_its_progress_bar_runs_resident(capfd):
    """``BayesGPR.fit`` defaults to ``progress=True`` (``bask/bayesgpr.py:550-564``): the run stays on the device, the bar
    follows it through the plan's segments (``bgp_mcmc_progress``), and the chain is the host-driven chain."""
    import bayes_skopt_amd as bask

    X, y = _data(300, 3, 4)
    out = []
    for resident in (True, False):
        gp = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1, 2]), random_state=2, normalize_y=True,
                           resident_sampler=resident)
        gp.fit(X, y, n_desired_samples=20 * 40, n_walkers_per_thread=20)  # (everything else the reference's defaults)
        out.append((gp.chain_.copy(), getattr(gp._sampler, "resident_runs", 0)))
    assert out[0][1] == 1 and out[1][1] == 0
    assert np.array_equal(out[0][0], out[1][0])
    err = capfd.readouterr().err
    assert "50/50" in err, err[-600:]  # tqdm's bar reached the end of the 40 + 10 burn-in steps (twice)
    # the progress mark itself: complete and monotone
    from bayes_skopt_amd import _lib  # noqa: F401
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1, 2]), random_state=2, normalize_y=True)
    gp.fit(X, y, n_desired_samples=40, n_burnin=1, n_walkers_per_thread=20, progress=False)
    ctx, seen = gp._ctx, []
    real = ctx.mcmc_progress

    def spy():
        v = real()
        seen.append(v)
        return v

    ctx.mcmc_progress = spy
    gp.sample(n_desired_samples=20 * 200, n_burnin=0, n_walkers_per_thread=20, progress=True)
    assert gp._sampler.resident_runs == 1 and seen and seen[-1] == 200 and all(b >= a for a, b in zip(seen, seen[1:]))


def test_resident_run_with_warped_inputs_replays_the_host_driven_chain():
    """``warp_inputs=True`` (``bask/bayesgpr.py:353-365``): the walkers' last 2 d entries are their own Beta-CDF parameters, the
    default warp priors (``scipy.stats.norm(0, 0.3).logpdf``, ``:463-466``) are the device's prior kind 3, the warped Gram build
    reads the parameters the step kernel wrote.  Positions bit for bit, log-probabilities at 1e-12."""
    import bayes_skopt_amd as bask

    for n, d, W, steps in ((60, 2, 20, 20), (300, 3, 32, 8)):
        X, y = _data(n, d, 4)
        out = []
        for resident in (False, True):
            gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=7, warp_inputs=True,
                               normalize_y=True, resident_sampler=resident)
            gp.fit(X, y, n_desired_samples=W * steps, n_burnin=0, n_walkers_per_thread=W, progress=False)
            s = gp._sampler
            out.append((s.get_chain().copy(), s.get_log_prob().copy(), getattr(s, "resident_runs", 0), gp.theta.copy(),
                        gp.warp_alphas_.copy(), s.naccepted.copy()))
        assert out[0][2] == 0 and out[1][2] == 1
        assert out[0][0].shape == (steps, W, d + 2 + 2 * d)
        assert np.array_equal(out[0][0], out[1][0]), "max position difference %.3e" % np.abs(out[0][0] - out[1][0]).max()
        np.testing.assert_allclose(out[1][1], out[0][1], rtol=1e-12, atol=0)
        assert np.array_equal(out[0][3], out[1][3]) and np.array_equal(out[0][4], out[1][4]) and np.array_equal(out[0][5], out[1][5])


def test_normal_priors_on_the_kernel_entries_run_resident():
    """A frozen ``scipy.stats.norm(...).logpdf`` is a device prior on any entry (kind 3: scipy's arithmetic operation by
    operation): same chain as the host-driven loop, log-probabilities EQUAL (no exp in this family)."""
    import scipy.stats as st

    pri = [st.norm(0.0, 2.0).logpdf, st.norm(-1.0, 1.5).logpdf, st.norm(-1.0, 1.5).logpdf, st.norm(loc=-4.0, scale=3.0).logpdf]
    _, s0 = _sample(200, 2, 20, 10, resident=False, priors=pri)
    _, s1 = _sample(200, 2, 20, 10, resident=True, priors=pri)
    assert getattr(s0, "resident_runs", 0) == 0 and s1.resident_runs == 1
    assert np.array_equal(s0.get_chain(), s1.get_chain())
    assert np.array_equal(s0.get_log_prob(), s1.get_log_prob())


@pytest.mark.parametrize("n,d,walkers,steps", [(200, 32, 256, 3), (120, 70, 160, 3)])
def test_ensembles_beyond_the_step_kernels_lds_run_resident(n, d, walkers, steps):
    """256 walkers at d = 32 (BASELINE config D's dimension: 18 048 doubles of ensemble state) fit the 160 KB form of the step
    kernel; d = 70 (p = 72: 23 360 doubles) takes the HBM form -- and, n <= 128 with more than 64 entries per walker, the step
    kernel instead of the fused one-launch half-step.  Same chain as the host-driven loop either way."""
    _, s0 = _sample(n, d, walkers, steps, resident=False)
    _, s1 = _sample(n, d, walkers, steps, resident=True)
    assert getattr(s0, "resident_runs", 0) == 0 and s1.resident_runs == 1
    assert np.array_equal(s0.get_chain(), s1.get_chain())
    np.testing.assert_allclose(s1.get_log_prob(), s0.get_log_prob(), rtol=1e-12, atol=0)
    assert np.array_equal(s0.naccepted, s1.naccepted)


def test_a_non_finite_proposal_raises_the_host_loops_error_and_leaves_its_generator_state():
    """emcee raises "At least one parameter value was infinite" / "... NaN" from the half-step that proposed the value; the
    resident run learns of it at the end, raises the same message and puts the generator where the host-driven loop leaves it."""
    import bayes_skopt_amd as bask
    from bayes_skopt_amd.bayesgpr import _AsyncLogProb
    from bayes_skopt_amd.sampler import EnsembleSampler
    from bayes_skopt_amd.utils import guess_priors
    from sklearn.gaussian_process.kernels import WhiteKernel

    X, y = _data(200, 2, 4)
    kernel = bask.construct_default_kernel([0, 1]) + WhiteKernel(1e-2)
    seen = set()
    for bad in (np.inf, np.nan):
        for walker in (4, 5, 11):
            got = []
            for resident in (False, True):
                gp = bask.BayesGPR(kernel=kernel, random_state=3, resident_sampler=resident)
                gp.kernel_ = kernel.clone_with_theta(kernel.theta)
                gp.X_train_, gp.y_train_ = X, (y - y.mean()) / y.std()
                gp._ensure_context(batch_hint=10)
                pos = np.tile(gp.kernel_.theta, (20, 1)) + 1e-2 * np.random.RandomState(0).randn(20, 4)
                pos[walker, 1] = bad  # (handed over with its log-probabilities: the initial-state check does not see it)
                s = EnsembleSampler(20, 4, _AsyncLogProb(gp), kwargs=dict(priors=guess_priors(gp.kernel_), warp_priors=None))
                s.random_state = np.random.RandomState(9).get_state()
                with pytest.raises(ValueError) as exc:
                    s.run_mcmc(pos, 6, skip_initial_state_check=True, log_prob0=np.zeros(20))
                assert s.iteration == 0 and s._chain is None
                got.append((str(exc.value), s._random.get_state()))
            (m0, a), (m1, b) = got
            assert m0 == m1 and "parameter value was" in m0, (m0, m1)
            assert np.array_equal(a[1], b[1]) and a[2:] == b[2:], (bad, walker)
            seen.add(m0)
    assert len(seen) == 2  # both of emcee's messages occurred


def _loopback_run(world, n, d, W, steps, persist, warp=False, seed=3):
    """One resident run of the same plan on ONE context, and on `world` contexts that share every half-step's proposal block
    through a loop-back communicator (threads of this process on the one GPU)."""
    import threading

    from bayes_skopt_amd import _lib
    from bayes_skopt_amd.bayesgpr import _device_prior
    from bayes_skopt_amd.sampler import EnsembleSampler
    from bayes_skopt_amd.utils import guess_priors
    import bayes_skopt_amd as bask
    from sklearn.gaussian_process.kernels import WhiteKernel
    import scipy.stats as st

    X, y = _data(n, d, 4)
    y = (y - y.mean()) / y.std()
    kernel = bask.construct_default_kernel(list(range(d))) + WhiteKernel(1e-2)
    table = [_device_prior(f) for f in guess_priors(kernel)]
    nwarp = 2 * d if warp else 0
    if warp:
        table += [_device_prior(st.norm(loc=0.0, scale=0.3).logpdf)] * nwarp
    p = d + 2 + nwarp
    kind = np.array([t[0] for t in table], dtype=np.int32)
    par = np.array([t[1] for t in table], dtype=np.float64)
    src, fixed = np.arange(d + 2), np.zeros(d + 2)
    rng = np.random.RandomState(seed)
    theta0 = np.concatenate([kernel.theta, np.zeros(nwarp)])
    coords = theta0 + 1e-2 * rng.randn(W, p)
    Ns = W // 2
    planner = EnsembleSampler(W, p, lambda T: np.zeros(len(T)))
    planner.random_state = np.random.RandomState(seed + 1).get_state()
    rows = list(planner._half_step_plans(steps))
    plan = (np.array([r[0] for r in rows]), np.array([r[1] for r in rows]), np.array([r[2][:, 0] for r in rows]),
            np.array([r[3] for r in rows]), np.log(np.random.RandomState(seed + 2).rand(2 * steps, Ns)))

    def make_ctx(max_batch):
        c = _lib.Context(X, y, 1e-10, max_batch=max_batch)
        c.set_persist(persist)
        return c

    ctx = make_ctx(Ns)
    H0 = coords[:, : d + 2]
    logp0 = (ctx.lml_warped(H0, coords[:, d + 2:]) if warp else ctx.lml(H0)) + 0.0
    ref = ctx.mcmc_run(coords, logp0, plan, src, fixed, kind, par, nwarp=nwarp)
    ctx.close()
    key = int(np.random.randint(1, 2**31))
    results, errors = [None] * world, []

    def rank_main(r):
        try:
            c = make_ctx(-(-Ns // world))
            comm = _lib.Comm.loopback(0, r, world, key)
            try:
                results[r] = c.mcmc_run(coords, logp0, plan, src, fixed, kind, par, comm=comm, nwarp=nwarp)
            finally:
                c.close()
                comm.close()
        except BaseException as exc:  # noqa: BLE001
            errors.append((r, exc))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    return ref, results


@pytest.mark.parametrize("world,n,d,W,steps,persist,warp", [
    (2, 300, 3, 40, 10, 0, False),      # two ranks, 10 rows each, launch schedule
    (3, 300, 3, 50, 6, 0, False),       # 25 proposals over three ranks: 8 / 8 / 9 rows
    (2, 100, 2, 20, 12, 0, False),      # n <= 128: the sharded run takes the step kernel + the fused LML kernel
    (4, 260, 2, 12, 5, 0, False),       # 6 proposals over four ranks: 1 / 2 / 1 / 2 rows
    (8, 200, 2, 12, 4, 0, False),       # more ranks than some shares have rows: 0 / 1 / 1 / 1 / 0 / 1 / 1 / 1
    (2, 200, 2, 24, 6, 0, True),        # warped walkers, sharded
])
def test_sharded_resident_run_over_a_loop_back_group_equals_the_single_context_run(world, n, d, W, steps, persist, warp):
    """The row-sharding logic of the multi-GPU resident sampler with world > 1 semantics on ONE GPU: `world` contexts (one host
    thread each) run the replicated step kernel, factorise their own rows of every half-step and exchange the log-likelihoods
    through a loop-back communicator (device copies + events in place of RCCL) on their streams.  Every rank's chain, final
    ensemble, log-probabilities and accept counts equal the single-context run's bit for bit."""
    ref, results = _loopback_run(world, n, d, W, steps, persist, warp)
    for r, got in enumerate(results):
        for name, a, b in zip(("chain", "logp", "coords", "logp_out", "naccepted"), ref, got):
            assert np.array_equal(a, b), (r, name)
        assert got[5][0] == 0 and got[5][1] == 0


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


_LOOP_CHILD = r"""
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from test_gpu_resident import _loopback_run
ref, results = _loopback_run(2, 1100, 4, 32, 4, 1, False)
same = all(np.array_equal(a, b) for got in results for a, b in zip(ref[:5], got[:5]))
print("RESULT " + json.dumps({"same": bool(same), "redone": [int(got[5][1]) for got in results], "ref_redone": int(ref[5][1])}))
"""


def test_a_launch_free_time_out_on_a_rank_of_a_sharded_run_makes_every_rank_redo_it():
    """Two loop-back ranks with the launch-free factorisation forced on and a wait bound no wait can meet: each rank's pack kernel
    reads its kernel's error word on the device and sends "redo" as its status word; EVERY rank's step kernel sees the words of
    all ranks, and at the end every rank redoes the whole run on the launch schedule (the same number of collectives again) --
    chains identical to the single-context run, `info[1] == 1` everywhere."""
    res = subprocess.run([sys.executable, "-c", _LOOP_CHILD % (ROOT, os.path.join(ROOT, "tests"))],
                         env=dict(os.environ, BGP_PS_TIMEOUT_TICKS="200"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    # (the single-context reference run met its time-out in the start ensemble's evaluation already and sits in the cool-down)
    assert d["same"] and d["redone"] == [1, 1], d
    assert "timed out" in res.stderr


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
