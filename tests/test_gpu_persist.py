"""Launch-free factorisation of small batches (bgp_set_persist / BGP_PERSIST; csrc/bgp_syrk4.hip::ps_kernel): ONE persistent
kernel per batch -- chain workgroups (one per matrix, or a pair per matrix with BGP_PS_PAIR=1; csrc/bgp_pf.h) first in the grid,
ticket-ordered left-looking tile workers behind them, one LDS array for both roles, device-scope flags between them.  The same
arithmetic in the same order as the multi-launch schedule, so the log-likelihoods, the failure statuses, the factors and a
whole MCMC chain must be BIT-identical to it; every in-kernel wait is bounded, a time-out falls back to the multi-launch
path with the right answer, and the context tries the launch-free path again after a cool-down (three time-outs switch it off)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_CHILD = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
out = {}
for n, d, B in %r:
    rng = np.random.RandomState(n + B)
    X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
    ad = np.full(n, 1e-10)
    if B > 3:
        X[5] = X[4]            # coinciding points (tiny pivots) ...
    if B > 3 and n > 300:
        X[200] = X[100]        # ... and a coinciding pair across two diagonal blocks whose second point carries a NEGATIVE
        ad[200] = -1e-3        # diagonal term: a walker with (almost) no noise fails at pivot 201, in block column 1
    ctx = _lib.Context(X, y, ad, max_batch=B)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.1 * rng.randn(B, d + 2)
    if B > 3:
        H[1, d + 1] = -np.inf if n <= 300 else np.log(0.02)  # no noise on one walker (where that still factorises), and
        H[2, 0] = np.nan       # a walker whose matrix is not a number: its factorisation must fail at the first pivot
    if B > 3 and n > 300:
        H[3, d + 1] = np.log(1e-6)  # the walker that fails in the middle of the matrix
    vals = []
    for rep in range(3):
        v, st = ctx.lml(H, return_status=True)
        vals.append([float(x).hex() for x in v])
    L, z = ctx.debug_workspace(0)
    out["%%d_%%d_%%d" %% (n, d, B)] = {"lml": vals, "status": st.tolist(), "Lsum": float(np.tril(L).sum()).hex(), "zsum": float(z.sum()).hex()}
    ctx.close()
print("RESULT " + json.dumps(out))
"""

SHAPES = [(300, 3, 8), (1024, 8, 32), (975, 8, 13), (640, 5, 1), (1100, 6, 50), (2048, 16, 9), (200, 3, 5)]


def _run(env, shapes=SHAPES):
    res = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, shapes)], env=dict(os.environ, **env), capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:]), res.stderr


@pytest.mark.parametrize("chains", ["auto", "single", "pair"])
def test_launch_free_factorisation_is_bit_identical_to_the_launch_schedule(chains):
    """"single": ONE chain workgroup per matrix (BGP_PS_PAIR=0); "pair": chain PAIRS (BGP_PS_PAIR=1: two chain workgroups per
    matrix alternate over the block columns, the idle one preparing the next diagonal block under the other's factorisation
    from the rows of W as they are published; the critical pre-updates in quadrants on a pool of their own); "auto": whichever
    bgp_pair_auto_rule picks per shape.  BGP_PS_PAIR is the only schedule switch the launch-free path still reads (round 5
    removed BGP_PS_NCRIT / BGP_PS_PSPLIT / BGP_PS_STREAM with the variants they selected)."""
    ref, _ = _run({"BGP_PERSIST": "0"})
    env = {"BGP_PERSIST": "1"}
    if chains != "auto":
        env["BGP_PS_PAIR"] = "1" if chains == "pair" else "0"
    got, err = _run(env)
    assert "timed out" not in err, err[-1500:]
    for k in ref:
        assert got[k]["lml"] == ref[k]["lml"], k        # every call, every matrix: the same bits
        assert got[k]["lml"][0] == got[k]["lml"][2], k  # and reproducible
        assert got[k]["status"] == ref[k]["status"], k
        assert got[k]["Lsum"] == ref[k]["Lsum"] and got[k]["zsum"] == ref[k]["zsum"], k  # the factor and z themselves
    st = got["1024_8_32"]["status"]
    # the NaN walker failed at pivot 1, the noise-free one at the second point of the coinciding pair, nobody else
    assert st[2] == 1 and st[3] == 201 and all(s == 0 for i, s in enumerate(st) if i not in (2, 3))
    assert got["1024_8_32"]["lml"][0][3] == float("-inf").hex()
    assert got["1024_8_32"]["lml"][0][2] == float("-inf").hex()


@pytest.mark.parametrize("gen", ["0", "1"])
def test_gram_blocks_built_inside_the_launch_free_kernel_keep_the_bits(gen):
    """BGP_PS_GEN=1: the tile workers generate the Gram blocks (and the working right-hand side) at the head of their ticket list
    instead of a Gram kernel in front of the launch; =0: the kernel in front, whatever the shape.  Shapes with failing matrices
    (a NaN walker, a pivot that fails in block column 1), ragged n, one matrix, a context-level diagonal vector."""
    shapes = [(300, 3, 8), (1024, 8, 32), (975, 8, 13), (640, 5, 1), (2048, 16, 9), (1100, 6, 24), (520, 2, 32)]
    ref, _ = _run({"BGP_PERSIST": "0"}, shapes)
    got, err = _run({"BGP_PERSIST": "1", "BGP_PS_PAIR": "0", "BGP_PS_GEN": gen}, shapes)
    assert "timed out" not in err and "not one this library reads" not in err, err[-1500:]
    for k in ref:
        assert got[k]["lml"] == ref[k]["lml"] and got[k]["status"] == ref[k]["status"], k
        assert got[k]["lml"][0] == got[k]["lml"][2], k
        assert got[k]["Lsum"] == ref[k]["Lsum"] and got[k]["zsum"] == ref[k]["zsum"], k


@pytest.mark.parametrize("pair", ["0", "1"])
def test_both_chain_schemes_keep_the_bits_on_further_shapes(pair):
    """The shapes the automatic rule sends to the OTHER scheme too: few and many matrices, 5 to 16 block columns, a ragged n."""
    shapes = [(1024, 8, 5), (2048, 16, 2), (640, 5, 3), (1536, 4, 9)]
    ref, _ = _run({"BGP_PERSIST": "0"}, shapes)
    got, err = _run({"BGP_PERSIST": "1", "BGP_PS_PAIR": pair}, shapes)
    assert "timed out" not in err, err[-1500:]
    for k in ref:
        assert got[k]["lml"] == ref[k]["lml"] and got[k]["status"] == ref[k]["status"], k
        assert got[k]["Lsum"] == ref[k]["Lsum"] and got[k]["zsum"] == ref[k]["zsum"], k


def test_a_wait_that_times_out_falls_back_with_the_right_answer():
    """BGP_PS_TIMEOUT_MS bounds every in-kernel wait.  With an absurdly small bound the first waits give up, both
    kernels drain, the host says so once and redoes the batch on the multi-launch path: same bits, no hang."""
    shapes = [(1024, 8, 16)]
    ref, _ = _run({"BGP_PERSIST": "0"}, shapes)
    # 100 MHz wall clock: a bound far below one potrf (27 us) cannot be met; the limit is given in ms -> use the debug
    # knob BGP_PS_TIMEOUT_TICKS
    got, err = _run({"BGP_PERSIST": "1", "BGP_PS_TIMEOUT_TICKS": "200"}, shapes)
    assert err.count("timed out") == 1, err[-1500:]
    assert got["1024_8_16"]["lml"] == ref["1024_8_16"]["lml"] and got["1024_8_16"]["status"] == ref["1024_8_16"]["status"]


def test_mcmc_chain_is_unchanged_by_the_launch_free_path():
    code = (
        "import sys, json; sys.path.insert(0, %r); import numpy as np; import bayes_skopt_amd as bask;"
        "rng = np.random.RandomState(3); X = rng.uniform(size=(700, 4)); y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(700);"
        "gp = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1, 2, 3]), random_state=5, normalize_y=True);"
        "gp.fit(X, y, n_desired_samples=120, n_burnin=3, n_walkers_per_thread=24, progress=False);"
        "print('RESULT ' + json.dumps([float(v).hex() for v in gp.chain_.ravel()] + [float(gp.log_marginal_likelihood_value_).hex()]))"
    ) % ROOT
    outs = []
    for env in ({"BGP_PERSIST": "0"}, {"BGP_PERSIST": "1"}):
        res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0 and "timed out" not in res.stderr, res.stderr[-2000:]
        outs.append([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1])
    assert outs[0] == outs[1]


_SY_CHILD = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
rng = np.random.RandomState(0)
n, d, m = 300, 3, 1500
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=4)
h = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
ctx.posterior(h[None, :])
hk = h.copy(); hk[-1] = -np.inf
Xq = rng.uniform(size=(m, d)); z = rng.randn(3, m)
outs = [ctx.sample_y(0, hk, Xq, z, jitter=1e-8) for _ in range(3)]
print("RESULT " + json.dumps([[float(v).hex() for v in o.ravel()[::97]] + [float(o.sum()).hex()] for o in outs]))
"""


def test_sample_y_covariance_on_the_launch_free_path_and_its_fallback():
    """The m x m predictive covariance of sample_y (one matrix, 12 block columns here, 79 at config E's 10 000 candidates)
    factorised launch-free: same draws, bit for bit; and when a wait times out the covariance is rebuilt and factorised by
    launches -- same draws again, one warning."""
    def run(env):
        res = subprocess.run([sys.executable, "-c", _SY_CHILD % ROOT], env=dict(os.environ, **env), capture_output=True, text=True,
                             timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:]), res.stderr

    ref, _ = run({"BGP_PERSIST": "0"})
    got, err = run({"BGP_PERSIST": "1"})
    assert "timed out" not in err and got == ref
    got, err = run({"BGP_PERSIST": "1", "BGP_PS_TIMEOUT_TICKS": "200"})
    assert err.count("timed out") == 1 and got == ref, err[-1500:]


def test_launch_free_path_against_the_sklearn_goldens():
    """The launch-free factorisation forced on (bgp_set_persist(1)) for every golden case with at least two block columns:
    ragged n with a per-point diagonal (M1-M4), BASELINE configs B / C / D at full size, a matrix that is singular at its
    second pivot -- log-likelihoods against scikit-learn 1.7.2 at 1e-6, statuses as the reference's LinAlgError cases."""
    import bayes_skopt_amd  # noqa: F401
    from bayes_skopt_amd import _lib
    from conftest import load_golden, synth

    g = load_golden("lml_sizes.npz")
    seen = 0
    for tag in ("M1", "M2", "M3", "M4", "B", "C", "D"):
        n, d, seed = [int(v) for v in g[tag + "_nd_seed"]]
        if n <= 128:
            continue
        X, y = synth(n, d, seed)
        ad = g[tag + "_alpha_diag"] if (tag + "_alpha_diag") in g.files else 1e-10
        ctx = _lib.Context(X, y, ad, max_batch=8)
        ctx.set_persist(1)
        got, status = ctx.lml(g[tag + "_theta"], return_status=True)
        assert np.all(status == 0), tag
        np.testing.assert_allclose(got, g[tag + "_lml"], rtol=1e-6, err_msg=tag)
        ctx.set_persist(0)
        np.testing.assert_array_equal(ctx.lml(g[tag + "_theta"]), got)  # and the launch schedule: the same bits
        ctx.close()
        seen += 1
    assert seen >= 5
    # not positive definite at a pivot inside block column 1: point 200 is a copy of point 150 and carries a negative
    # diagonal term (LinAlgError in the reference: -inf, sklearn/_gpr.py:588-589); with enough noise the matrix is fine
    rng = np.random.RandomState(4)
    X = rng.uniform(size=(300, 2))
    X[200] = X[150]
    yv = rng.randn(300)
    ad = np.full(300, 1e-10)
    ad[200] = -1e-3
    ctx = _lib.Context(X, yv, ad, max_batch=4)
    ctx.set_persist(1)
    H = np.array([[0.0, np.log(0.3), np.log(0.3), np.log(1e-6)], [0.0, np.log(0.3), np.log(0.3), np.log(0.1)]])
    got, status = ctx.lml(H, return_status=True)
    assert got[0] == -np.inf and status[0] == 201 and status[1] == 0 and np.isfinite(got[1])
    ctx.close()


def test_warped_batches_take_the_launch_free_path_with_the_same_bits():
    """Per-walker input warping: the Gram build works on per-walker inputs, the factorisation behind it is
    the same -- forced launch-free and on the launch schedule the log-likelihoods are the same bits, and both agree with
    the oracle (Beta-CDF warp + scikit-learn's formula) at 1e-6; the asynchronous submit / wait pair too."""
    import bayes_skopt_amd  # noqa: F401
    from bayes_skopt_amd import _lib
    from oracle import gp_oracle as O

    rng = np.random.RandomState(12)
    n, d, B = 700, 3, 6
    X = rng.uniform(0.02, 0.98, size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    y = (y - y.mean()) / y.std()
    H = np.array([0.0, -1.0, -1.1, -0.9, -3.5]) + 0.1 * rng.randn(B, d + 2)
    Wp = 0.3 * rng.randn(B, 2 * d)
    ctx = _lib.Context(X, y, 1e-10, max_batch=B)
    out = {}
    for mode in (0, 1):
        ctx.set_persist(mode)
        out[mode] = ctx.lml_warped(H, Wp)
        assert ctx.lml_warped_submit(H, Wp)
        np.testing.assert_array_equal(ctx.lml_wait(), out[mode])
    np.testing.assert_array_equal(out[0], out[1])
    for b in (0, B - 1):
        np.testing.assert_allclose(out[1][b], O.lml_warped(X, y, np.full(n, 1e-10), H[b], Wp[b]), rtol=1e-6)
    ctx.close()


def test_the_automatic_choice_takes_the_launch_free_path_where_it_measured_faster():
    """bgp_persist_auto_rule (at least 6 block columns, matrices x block columns <= 400, >= 100 unless there are 12 block
    columns -- or bgp_pair_auto_rule: few matrices, where the chain pairs win from 3 block columns on): seen from outside
    through the in-kernel trace, which only a launch-free call writes."""
    code = (
        "import sys, json; sys.path.insert(0, %r); import numpy as np; import bayes_skopt_amd; from bayes_skopt_amd import _lib\n"
        "out = {}\n"
        "for n, d, B in ((1024, 4, 32), (1024, 4, 4), (512, 4, 32), (512, 4, 4), (1536, 4, 2), (1024, 4, 64), (768, 4, 32), (256, 4, 4)):\n"
        "    rng = np.random.RandomState(1); X = rng.uniform(size=(n, d)); y = rng.randn(n)\n"
        "    ctx = _lib.Context(X, y, 1e-6, max_batch=B)\n"
        "    H = np.array([0.0] + [-1.0] * d + [-2.0]) + 0.05 * rng.randn(B, d + 2)\n"
        "    ctx.lml(H)\n"
        "    out['%%d_%%d' %% (n, B)] = ctx.ps_trace() is not None\n"
        "    ctx.close()\n"
        "print('RESULT ' + json.dumps(out))\n"
    ) % ROOT
    env = {k: v for k, v in os.environ.items() if k != "BGP_PERSIST"}
    env["BGP_PS_TRACE"] = "1"
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    got = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert got == {"1024_32": True, "1024_4": True, "512_32": False, "512_4": True, "1536_2": True, "1024_64": False, "768_32": True,
                   "256_4": False}, got


def test_time_out_policy_cool_down_then_off_for_good():
    """One transient time-out must not cost a context the path for life (and three must): with every wait timing out and a
    cool-down of 2 eligible calls the context goes launch-free, by launches twice, launch-free again, ... and after the third
    time-out stays on the launches; bgp_persist_stats counts; bgp_set_persist(ctx, 1) re-arms.  Results are right throughout."""
    code = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np
import bayes_skopt_amd
from bayes_skopt_amd import _lib
rng = np.random.RandomState(1)
n, d, B = 1024, 4, 8
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1))
ctx = _lib.Context(X, y, 1e-10, max_batch=B)
H = np.concatenate([[0.0], np.full(d, np.log(0.4)), [np.log(0.02)]]) + 0.1 * rng.randn(B, d + 2)
ctx.set_persist(0); ref = ctx.lml(H)
ctx.set_persist(1)
log = []
for i in range(12):
    ok = bool(np.array_equal(ctx.lml(H), ref))
    s = ctx.persist_stats(); log.append([ok, s["calls"], s["timeouts"], s["disabled"], s["cooldown_left"]])
ctx.set_persist(1)
ok = bool(np.array_equal(ctx.lml(H), ref)); s = ctx.persist_stats(); log.append([ok, s["calls"], s["timeouts"], s["disabled"], s["cooldown_left"]])
print("RESULT " + json.dumps(log))
""" % ROOT
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BGP_PS_TIMEOUT_TICKS="200", BGP_PS_COOLDOWN="2"),
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    log = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert all(r[0] for r in log)
    # call 1: launch-free, times out (1), two calls by launches, call 4 launch-free again (2), two by launches, call 7 (3): off
    assert [r[1:3] for r in log[:12]] == [[1, 1], [1, 1], [1, 1], [2, 2], [2, 2], [2, 2], [3, 3]] + [[3, 3]] * 5
    assert log[6][3] is True and log[6][4] == 0 and log[11][3] is True
    assert log[12][1:3] == [4, 4]  # re-armed by bgp_set_persist(ctx, 1): tried (and timed out) once more
    assert res.stderr.count("timed out") == 3  # said so three times, then quietly
