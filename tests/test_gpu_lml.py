"""GPU parity of the LML hot path (kernel build -> blocked Cholesky -> fused solve -> LML) through the
C-ABI against the golden vectors (sklearn 1.7.2) and the oracle.  Tolerance: 1e-6 relative
(BASELINE.json north_star); observed errors are reported by tools/archive/gpu_probe.py."""
import numpy as np
import pytest

from conftest import load_golden, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-6


@pytest.fixture(scope="module")
def lib():
    import bayes_skopt_amd  # noqa: F401
    from bayes_skopt_amd import _lib

    assert _lib.device_count() >= 1, "no HIP device: GPU tests need an MI355X"
    return _lib


def test_mfma_f64_fragment_layout(lib):
    rows, cols = lib.mfma_f64_layout()
    lane = np.arange(64)[:, None]
    reg = np.arange(4)[None, :]
    np.testing.assert_array_equal(cols, np.broadcast_to(lane & 15, (64, 4)))
    np.testing.assert_array_equal(rows, (lane >> 4) + 4 * reg)


def test_pivot_root_accuracy(lib):
    """The pivot chain's sqrt / 1/sqrt (bgp_pf.h::pf_pivot_root; LAPACK dpotrf takes a correctly rounded sqrt): within 1 ulp /
    2 ulp of the correctly rounded values on 2e6 arguments spanning 1e-12 .. 1e6 (measured: 0.5 / 1.5)."""
    x = 10.0 ** np.random.RandomState(0).uniform(-12.0, 6.0, size=2_000_000)
    s, r = lib.pivot_root(x)
    xl = x.astype(np.longdouble)
    rs = np.sqrt(xl)
    ri = 1.0 / rs
    ulp_s = np.spacing(rs.astype(np.float64))
    ulp_i = np.spacing(ri.astype(np.float64))
    err_s = np.abs(s.astype(np.longdouble) - rs) / ulp_s
    err_i = np.abs(r.astype(np.longdouble) - ri) / ulp_i
    assert float(err_s.max()) <= 1.0 and float(err_i.max()) <= 2.0, (float(err_s.max()), float(err_i.max()))
    assert np.mean(s == rs.astype(np.float64)) > 0.99


def test_kernel_matrix_all_forms(lib):
    g = load_golden("lml_small.npz")
    for c in range(int(g["n_cases"])):
        pre = f"c{c}_"
        st, form = [str(s) for s in g[pre + "meta"]]
        ctx = lib.Context(g[pre + "X"], g[pre + "y"], g[pre + "alpha_diag"], form=form, stationary=st, max_batch=8)
        for b, th in enumerate(g[pre + "theta"][:2]):
            K = ctx.kernel_matrix(th)
            np.testing.assert_allclose(K, g[pre + "K"][b], rtol=1e-12, atol=1e-14)
        ctx.close()


def test_lml_small_all_forms(lib):
    g = load_golden("lml_small.npz")
    for c in range(int(g["n_cases"])):
        pre = f"c{c}_"
        st, form = [str(s) for s in g[pre + "meta"]]
        ctx = lib.Context(g[pre + "X"], g[pre + "y"], g[pre + "alpha_diag"], form=form, stationary=st, max_batch=4)
        got, status = ctx.lml(g[pre + "theta"], return_status=True)  # B=6 > max_batch: exercises chunking
        assert np.all(status == 0)
        np.testing.assert_allclose(got, g[pre + "lml"], rtol=RTOL)
        ctx.close()


def test_lml_config_A_and_edge_thetas(lib):
    g = load_golden("lml_sizes.npz")
    ctx = lib.Context(g["A_X"], g["A_y"], 1e-10, max_batch=16)
    got = ctx.lml(g["A_theta"])
    np.testing.assert_allclose(got, g["A_lml"], rtol=RTOL)
    ctx.close()


@pytest.mark.parametrize("tag", ["M1", "M2", "M3", "M4"])
def test_lml_ragged_sizes_vector_alpha(lib, tag):
    g = load_golden("lml_sizes.npz")
    n, d, seed = [int(v) for v in g[tag + "_nd_seed"]]
    X, y = synth(n, d, seed)
    ctx = lib.Context(X, y, g[tag + "_alpha_diag"], max_batch=8)
    got = ctx.lml(g[tag + "_theta"])
    np.testing.assert_allclose(got, g[tag + "_lml"], rtol=RTOL)
    ctx.close()


def test_singular_matrix_gives_minus_inf_and_status(lib):
    g = load_golden("lml_sizes.npz")
    ctx = lib.Context(g["S_X"], g["S_y"], np.zeros(16), max_batch=4)
    ok = np.array([0.0, np.log(0.3), np.log(0.3), np.log(0.1)])
    H = np.vstack([g["S_theta"][0], ok, g["S_theta"][0]])
    got, status = ctx.lml(H, return_status=True)
    assert got[0] == -np.inf and got[2] == -np.inf and np.isfinite(got[1])
    assert status[0] == 2 and status[2] == 2 and status[1] == 0  # second pivot is exactly zero
    ctx.close()


@pytest.mark.parametrize("tag", ["B", "C", "D"])
def test_lml_baseline_configs_full_size(lib, tag):
    g = load_golden("lml_sizes.npz")
    n, d, seed = [int(v) for v in g[tag + "_nd_seed"]]
    X, y = synth(n, d, seed)
    ctx = lib.Context(X, y, 1e-10, max_batch=4)
    got = ctx.lml(g[tag + "_theta"])
    np.testing.assert_allclose(got, g[tag + "_lml"], rtol=RTOL)
    ctx.close()


def test_update_data_grows_n(lib):
    """tell() appends points: same context, larger n (bask/optimizer.py:288-320)."""
    from oracle import gp_oracle as O

    X, y = synth(200, 3, 9)
    th = np.array([[0.1, -1.0, -1.2, -0.9, -4.0]])
    ctx = lib.Context(X[:100], y[:100], 1e-10, max_batch=2)
    for n in (100, 129, 200):
        if n != 100:
            ctx.update_data(X[:n], y[:n], np.full(n, 1e-10))
        got = ctx.lml(th)
        np.testing.assert_allclose(got[0], O.lml(X[:n], y[:n], np.full(n, 1e-10), th[0]), rtol=RTOL)
    ctx.close()


# ---- size-independent properties at the BASELINE full sizes (no CPU oracle needed at that size)
def test_full_size_permutation_invariance(lib):
    """LML(P X, P y) == LML(X, y): a row permutation changes every tile, every pivot order and every partial sum,
    the value only by rounding (n = 4096, d = 32: BASELINE config D)."""
    n, d = 4096, 32
    X, y = synth(n, d, 7)
    perm = np.random.RandomState(8).permutation(n)
    H = np.concatenate([[0.1], np.full(d, np.log(0.35)), [np.log(0.02)]])[None, :] + \
        0.05 * np.random.RandomState(9).randn(2, d + 2)
    a = lib.Context(X, y, np.full(n, 1e-10), max_batch=2)
    la = a.lml(H)
    a.close()
    b = lib.Context(X[perm], y[perm], np.full(n, 1e-10), max_batch=2)
    lb = b.lml(H)
    b.close()
    np.testing.assert_allclose(la, lb, rtol=1e-10)


def test_full_size_known_answer_white_dominated(lib):
    """Signal variance -> 0: K = (c + s2 + alpha) I up to 1e-18, so the LML has the closed form
    -y'y / (2 d0) - n/2 log d0 - n/2 log 2 pi at ANY size (n = 4096, a ragged n = 3001, and n = 12288: 96 block
    columns, a 1.2 GB matrix)."""
    for n, d in ((4096, 8), (3001, 3), (12288, 4)):
        X, y = synth(n, d, 11)
        c, s2, a0 = np.exp(-42.0), 4.0, 1e-3
        h = np.concatenate([[-42.0], np.full(d, np.log(0.3)), [np.log(s2)]])
        ctx = lib.Context(X, y, np.full(n, a0), max_batch=1)
        got = ctx.lml(h)[0]
        ctx.close()
        d0 = c + s2 + a0
        ref = -0.5 * float(y @ y) / d0 - 0.5 * n * np.log(d0) - 0.5 * n * np.log(2 * np.pi)
        np.testing.assert_allclose(got, ref, rtol=1e-12)


def test_full_size_block_diagonal_additivity(lib):
    """Two clusters further apart than the kernel's range: the cross-covariance underflows to exactly 0, K is block
    diagonal and LML(A u B) = LML(A) + LML(B) + n/2-terms -- a checksum-of-checksums over 2 x 1024 points whose
    blocks straddle the 128-row tiles differently in the joint and in the separate factorisations."""
    na, nb, d = 1000, 1100, 4
    rng = np.random.RandomState(21)
    XA = rng.uniform(0.0, 0.2, size=(na, d))
    XB = rng.uniform(0.8, 1.0, size=(nb, d))
    yA, yB = np.sin(20 * XA.sum(1)), np.cos(17 * XB.sum(1))
    h = np.concatenate([[0.2], np.full(d, np.log(2e-4)), [np.log(0.05)]])  # ell so small that exp(-sqrt5 r/ell) == 0 across
    def lml_of(X, y):
        ctx = lib.Context(X, y, np.full(len(y), 1e-10), max_batch=1)
        v = ctx.lml(h)[0]
        ctx.close()
        return v
    order = rng.permutation(na + nb)  # interleave the clusters: the block structure is hidden from the tiling
    joint = lml_of(np.vstack([XA, XB])[order], np.concatenate([yA, yB])[order])
    np.testing.assert_allclose(joint, lml_of(XA, yA) + lml_of(XB, yB), rtol=1e-12)


def test_submit_wait_equals_synchronous_call_and_guards_the_workspace(lib):
    """bgp_lml_batch_submit / _wait (the sampler's asynchronous form: proposals and results through pinned memory) give
    the bits of bgp_lml_batch; while a batch is pending the workspace is its own -- a second submit or a synchronous
    call is refused, and a wait without a submit too."""
    rng = np.random.RandomState(3)
    X = rng.uniform(size=(300, 3))
    y = np.sin(3 * X.sum(1))
    ctx = lib.Context(X, y, 1e-8, max_batch=8)
    H = np.concatenate([[0.0], np.full(3, np.log(0.4)), [np.log(0.02)]]) + 0.1 * rng.randn(8, 5)
    ref = ctx.lml(H)
    assert ctx.lml_submit(H)
    with pytest.raises(Exception, match="pending"):
        ctx.lml(H)
    with pytest.raises(Exception, match="pending"):
        ctx.posterior(H[:1])
    with pytest.raises(Exception, match="pending"):
        ctx.update_data(X, y, 1e-8)
    with pytest.raises(Exception, match="pending"):
        lib._check(lib.load().bgp_lml_batch_submit(ctx._h, 8, lib._p(H)), "submit")
    np.testing.assert_array_equal(ctx.lml_wait(), ref)
    ctx._pending = 8  # (python-side bookkeeping only: the library must notice that nothing was submitted)
    with pytest.raises(Exception, match="nothing submitted"):
        ctx.lml_wait()
    np.testing.assert_array_equal(ctx.lml(H), ref)
    ctx.close()
