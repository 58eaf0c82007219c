"""Edge cases and size-independent properties of the device LML path (through the C-ABI)."""
import numpy as np
import pytest

from conftest import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-6


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


@pytest.mark.parametrize("n,d", [(1, 1), (2, 1), (3, 2), (5, 7), (127, 3), (128, 1), (129, 2), (255, 40), (256, 33), (385, 17)])
def test_small_and_ragged_shapes(lib, O, n, d):
    X, y = synth(max(n, 4), d, 100 + n)
    X, y = X[:n], y[:n]
    H = np.concatenate([[0.1], np.full(d, np.log(0.6)), [np.log(0.05)]]) + 0.1 * np.random.RandomState(n).randn(3, d + 2)
    ad = np.full(n, 1e-10)
    ctx = lib.Context(X, y, ad, max_batch=2)  # B=3 > max_batch: chunked
    got = ctx.lml(H)
    ref = O.lml_batch(X, y, ad, H)
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=1e-9)
    ctx.close()


@pytest.mark.parametrize("B", [1, 7, 9, 17])
def test_batch_sizes_not_multiple_of_eight(lib, O, B):
    n, d = 200, 3
    X, y = synth(n, d, 55)
    H = np.array([0.0, -1.0, -1.2, -0.8, -3.5]) + 0.15 * np.random.RandomState(B).randn(B, d + 2)
    ctx = lib.Context(X, y, 1e-10, max_batch=32)
    got = ctx.lml(H)
    np.testing.assert_allclose(got, O.lml_batch(X, y, np.full(n, 1e-10), H), rtol=RTOL)
    ctx.close()


def test_mixed_failures_across_chunks(lib, O):
    """Non-PD items anywhere in a batch (and across max_batch chunk boundaries) only affect themselves."""
    n, d = 140, 2
    X, y = synth(n, d, 66)
    X[1] = X[0]  # duplicated point: exactly singular without jitter/noise (second pivot is exactly 0)
    ad = np.zeros(n)
    good = np.array([0.0, -1.0, -1.1, -3.0])
    bad = np.array([0.0, -1.0, -1.1, -np.inf])
    H = np.array([good, bad, good + 0.1, bad, bad, good - 0.1, good, bad, good + 0.2])
    ctx = lib.Context(X, y, ad, max_batch=4)
    got, status = ctx.lml(H, return_status=True)
    isbad = np.array([0, 1, 0, 1, 1, 0, 0, 1, 0], dtype=bool)
    assert np.all(got[isbad] == -np.inf) and np.all(status[isbad] == 2)  # pivot of the duplicate (1-based)
    assert np.all(status[~isbad] == 0)
    np.testing.assert_allclose(got[~isbad], O.lml_batch(X, y, ad, H[~isbad]), rtol=RTOL)
    ctx.close()


def test_overflowing_hyperparameters_give_minus_inf_not_nan(lib):
    n, d = 64, 2
    X, y = synth(n, d, 77)
    ctx = lib.Context(X, y, 1e-10, max_batch=4)
    H = np.array([[800.0, -1.0, -1.0, -3.0],     # c = inf
                  [0.0, -1.0, -1.0, 800.0],      # s2 = inf
                  [0.0, 800.0, 800.0, -800.0],   # l = inf, s2 = 0 -> rank one + 1e-10
                  [0.0, -1.0, -1.0, -3.0]])
    got, status = ctx.lml(H, return_status=True)
    assert not np.any(np.isnan(got))
    assert got[0] == -np.inf and got[1] == -np.inf and np.isfinite(got[3])
    assert status[0] != 0 and status[1] != 0 and status[3] == 0
    ctx.close()


def test_update_data_shrinks_and_grows(lib, O):
    X, y = synth(300, 2, 88)
    th = np.array([[0.0, -1.0, -1.2, -3.0]])
    ctx = lib.Context(X, y, 1e-10, max_batch=2)
    for n in (300, 40, 260, 129):
        ctx.update_data(X[:n], y[:n], np.full(n, 1e-10))
        np.testing.assert_allclose(ctx.lml(th)[0], O.lml(X[:n], y[:n], np.full(n, 1e-10), th[0]), rtol=RTOL)
    ctx.close()


@pytest.mark.parametrize("n,d", [(2048, 16), (4096, 32)])
def test_full_size_properties(lib, n, d):
    """Size-independent properties at the BASELINE sizes (the oracle is too slow to sweep there):
    (1) permuting the data rows leaves the LML unchanged; (2) with a dominant noise level the LML
    approaches the closed form of an isotropic Gaussian; (3) LML(theta) is reproducible bit-for-bit."""
    X, y = synth(n, d, 0)
    base = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
    H = base + 0.1 * np.random.RandomState(1).randn(2, d + 2)
    ctx = lib.Context(X, y, 1e-10, max_batch=4)
    a = ctx.lml(H)
    b = ctx.lml(H)
    np.testing.assert_array_equal(a, b)
    perm = np.random.RandomState(2).permutation(n)
    ctx2 = lib.Context(X[perm], y[perm], 1e-10, max_batch=4)
    np.testing.assert_allclose(ctx2.lml(H), a, rtol=1e-9)
    big = base.copy()
    big[-1] = 20.0  # s2 = e^20 >> c
    s2 = np.exp(20.0)
    closed = -0.5 * float(y @ y) / s2 - 0.5 * n * 20.0 - 0.5 * n * np.log(2 * np.pi)
    np.testing.assert_allclose(ctx.lml(big[None, :])[0], closed, rtol=1e-8)
    ctx.close()
    ctx2.close()


def test_predict_interpolates_smooth_training_points_with_tiny_noise(lib):
    n, d = 150, 2
    X, _ = synth(n, d, 99)
    y = np.sin(3.0 * X.sum(axis=1))  # noise-free targets: a near-interpolating GP must reproduce them
    th = np.array([0.0, -0.8, -0.8, np.log(1e-8)])
    ctx = lib.Context(X, y, 1e-10, max_batch=2)
    ctx.posterior(th)
    mean, var = ctx.predict(th, X[:50])
    np.testing.assert_allclose(mean[0], y[:50], atol=2e-4)
    assert np.all(var[0] < 1e-5)
    ctx.close()


def test_schedule_variants_agree_bitwise(lib, O):
    """Panel-group sizes (BGP_PANELS), walker-group stream counts, the unpipelined Gram build and the Gram generation
    fused into the trailing update only regroup the same operations: LML, alpha and K^-1 must come out with the same
    BITS as the default schedule."""
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import bayes_skopt_amd; from bayes_skopt_amd import _lib;"
        "rng=np.random.RandomState(5); n,d=700,6; X=rng.uniform(size=(n,d)); y=np.sin(3*X.sum(1));"
        "H=np.concatenate([[0.],np.full(d,np.log(.4)),[np.log(.02)]])+0.1*np.random.RandomState(6).randn(5,d+2);"
        "c=_lib.Context(X,y,1e-10,max_batch=8); l=c.lml(H).tolist(); r=c.posterior(H[:3], want_alpha=True, want_K_inv=True);"
        "print(repr(l + r['alpha'].sum(1).tolist() + [float(np.trace(k)) for k in r['K_inv']]))"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {"BGP_PANELS": "3"}, {"BGP_PANELS": "1"}, {"BGP_STREAMS": "1"}, {"BGP_STREAMS": "3"},
                {"BGP_PANELS": "4"}, {"BGP_PANELS": "16"}, {"BGP_STREAMS": "2", "BGP_PANELS": "2"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(np.array(eval(r.stdout.strip().splitlines()[-1])))
    for o in outs[1:]:  # (panel grouping and stream count only regroup the same operations)
        np.testing.assert_array_equal(o, outs[0])
    rng = np.random.RandomState(5)
    X = rng.uniform(size=(700, 6))
    y = np.sin(3 * X.sum(1))
    H = np.concatenate([[0.0], np.full(6, np.log(0.4)), [np.log(0.02)]]) + 0.1 * np.random.RandomState(6).randn(5, 8)
    # (each output: 5 LML values, then alpha sums and trace(K^-1) of three posterior builds on the ring kernels with the
    # active-row remap; round 1's kernels left the library in round 3 -- tools/legacy/ keeps them for the A/B benches)
    np.testing.assert_allclose(outs[0][:5], O.lml_batch(X, y, np.full(700, 1e-10), H), rtol=RTOL)


def test_bitwise_reproducible_and_batch_split_invariant():
    """Same inputs -> same bits, run after run and however a batch is split: no floating-point atomics on
    the LML / gradient / predict result paths (seeded runs of the reference are reproducible too)."""
    from bayes_skopt_amd import _lib

    for n, d in ((96, 2), (300, 3), (1024, 8)):
        rng = np.random.RandomState(n)
        X = rng.uniform(size=(n, d))
        y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
        ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=16)
        H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.1 * rng.randn(12, d + 2)
        a = ctx.lml(H)
        for _ in range(3):
            np.testing.assert_array_equal(ctx.lml(H), a)
        np.testing.assert_array_equal(np.concatenate([ctx.lml(H[:5]), ctx.lml(H[5:])]), a)
        np.testing.assert_array_equal(np.concatenate([ctx.lml(H[i:i + 1]) for i in range(12)]), a)
        l0, g0 = ctx.lml_grad(H[:3])[:2]
        for _ in range(3):
            l1, g1 = ctx.lml_grad(H[:3])[:2]
            np.testing.assert_array_equal(l1, l0)
            np.testing.assert_array_equal(g1, g0)
        ctx.posterior(H[:1])
        Xq = rng.uniform(size=(700, d))
        m0, v0 = ctx.predict(H[:1], Xq)
        for _ in range(3):
            m1, v1 = ctx.predict(H[:1], Xq)
            np.testing.assert_array_equal(m1, m0)
            np.testing.assert_array_equal(v1, v0)
        ctx.close()


def test_walker_group_streams_bit_identical():
    """Batches of >= 64 proposals are split over two walker-group streams by default (their launches overlap
    on the GPU); the result must not depend on the grouping."""
    from bayes_skopt_amd import _lib

    rng = np.random.RandomState(11)
    n, d, B = 260, 3, 80
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.2 * rng.randn(B, d + 2)
    H[7, -1] = np.log(1e-300)  # one proposal that may fail must not disturb its group
    ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
    auto = ctx.lml(H)
    for ns in (1, 2, 3):
        ctx.set_streams(ns)
        np.testing.assert_array_equal(ctx.lml(H), auto)
    ctx.close()


def test_pipelined_gram_build_matches_plain_build_and_oracle(lib, O):
    """Launches of >= 2048 tiles take the pipelined Gram build (xscale_kernel + kbuild2_kernel on the LDS-DMA ring,
    full-square and lower-triangular tile lists); smaller ones the plain kernel.  Same K bit for bit -- the log-likelihoods of
    16 matrices in ONE call (16 x 153 tiles: pipelined build) equal those of 16 calls of one matrix (8 x 153 tiles: plain build)
    -- and within 1e-12 of the oracle, every kernel family, ragged n, d > 16 (two k-blocks)."""
    import os
    import subprocess
    import sys

    n, d = 2100, 19
    rng = np.random.RandomState(3)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1))
    h = np.concatenate([[0.2], np.log(0.5) + 0.2 * rng.randn(d), [np.log(0.03)]])
    alpha = 10.0 ** rng.uniform(-8, -3, size=n)
    for stat, form in (("matern52", "product"), ("rbf", "sum"), ("matern12", "product")):
        ctx = lib.Context(X, y, alpha, form=form, stationary=stat, max_batch=2)
        K = ctx.kernel_matrix(h)  # full square: 8 x 17^2 tiles -> pipelined build
        Ko = O.gram_with_jitter(X, alpha, h, stationary=stat, form=form)
        np.testing.assert_allclose(K, Ko, rtol=1e-12, atol=1e-300)
        ctx.close()
    ctx = lib.Context(X, y, 1e-10, max_batch=16)
    ctx.set_persist(0)
    H = h + 0.05 * np.random.RandomState(4).randn(16, d + 2)
    together = ctx.lml(H)
    one_by_one = np.concatenate([ctx.lml(H[i : i + 1]) for i in range(16)])
    np.testing.assert_array_equal(together, one_by_one)
    np.testing.assert_allclose(together[:2], O.lml_batch(X, y, np.full(n, 1e-10), H[:2]), rtol=1e-9)
    ctx.close()


def test_unknown_switch_is_reported():
    """A/B measurements hang on the BGP_* switches: one the library does not read (a typo, a switch of an older round)
    must not pass silently as 'the variant'."""
    import os
    import subprocess
    import sys

    code = ("import sys, numpy as np; sys.path.insert(0, %r); import bayes_skopt_amd; from bayes_skopt_amd import _lib;"
            "c=_lib.Context(np.random.RandomState(0).rand(40,2), np.zeros(40), 1e-6, max_batch=2); print('ok')"
            ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BGP_SYRK4="1", BGP_PANELS="2"), capture_output=True,
                       text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]
    assert "BGP_SYRK4 is not one this library reads" in r.stderr and "BGP_PANELS" not in r.stderr


def test_allocation_failure_is_an_error_and_the_library_stays_usable(lib, O):
    """A workspace the device cannot hold (4 000 matrices of 4096^2 doubles = 537 GB of 288) is a BgpError naming the
    allocation -- no crash, no sticky HIP error: the next context works."""
    rng = np.random.RandomState(0)
    n, d = 4096, 3
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1))
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * rng.randn(8, d + 2)
    with pytest.raises(lib.BgpError) as err:
        lib.Context(X, y, 1e-10, max_batch=4000)  # (the workspace is allocated with the context)
    assert "hipMalloc" in str(err.value) and "out of memory" in str(err.value)
    X2, y2 = X[:300], y[:300]
    ctx = lib.Context(X2, y2, 1e-10, max_batch=4)
    np.testing.assert_allclose(ctx.lml(H[:4]), O.lml_batch(X2, y2, np.full(300, 1e-10), H[:4]), rtol=RTOL)
    ctx.close()


def test_two_contexts_from_two_threads_and_a_wait_on_another_thread(lib, O):
    """The pinned transfer arena is process-wide and keyed by stream: contexts driven from different threads at the same
    time, and a batch submitted on one thread and collected on another, give the right answers (every caller-buffer
    transfer -- chunked LML calls beyond max_batch, predict -- goes through the arena)."""
    import threading

    rng = np.random.RandomState(1)
    jobs = []
    for k, (n, d) in enumerate(((700, 3), (520, 5))):
        X = rng.uniform(size=(n, d))
        y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
        H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.1 * rng.randn(20, d + 2)
        jobs.append((X, y, H, O.lml_batch(X, y, np.full(n, 1e-10), H)))
    out, errs = {}, []

    def work(k):
        try:
            X, y, H, _ref = jobs[k]
            ctx = lib.Context(X, y, 1e-10, max_batch=8)  # 20 proposals in chunks of 8: the arena path
            vals = [ctx.lml(H) for _ in range(15)]
            ctx.posterior(H[:1])
            mean, var = ctx.predict(H[:1], X[:64])
            out[k] = (vals, mean, var)
            ctx.close()
        except Exception as exc:  # pragma: no cover
            errs.append(exc)

    ts = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for k in (0, 1):
        for v in out[k][0]:
            np.testing.assert_allclose(v, jobs[k][3], rtol=RTOL)
    # submit here, wait there
    X, y, H, ref = jobs[0]
    ctx = lib.Context(X, y, 1e-10, max_batch=32)
    got = {}
    assert ctx.lml_submit(H)
    t = threading.Thread(target=lambda: got.setdefault("v", ctx.lml_wait()))
    t.start()
    t.join()
    np.testing.assert_allclose(got["v"], ref, rtol=RTOL)
    ctx.close()


def test_a_failing_call_on_one_thread_leaves_another_threads_downloads_alone(lib, O):
    """The arena's bookkeeping is shared, a FAILED call must not be: thread A runs into the routine BGP_ERR_NOTPD of
    bgp_sample_y (a singular predictive covariance, no jitter) over and over while thread B's chunked LML calls and predicts
    have their downloads staged in the same arena.  Dropping "all pending downloads" on A's failure (rounds 3-4) left B's
    results unpacked -- uninitialised caller buffers under BGP_OK; the drop is scoped to the failing thread now."""
    import threading

    rng = np.random.RandomState(4)
    n, d = 300, 2
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
    H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.1 * rng.randn(12, d + 2)
    ref = O.lml_batch(X, y, np.full(n, 1e-10), H)
    Xq = rng.uniform(size=(40, d))
    stop, errs, counts = threading.Event(), [], {"notpd": 0, "ok": 0}

    def failing():
        try:
            ctx = lib.Context(X, y, 1e-10, max_batch=2)
            ctx.posterior(H[:1])
            hk = H[0].copy()
            hk[-1] = -np.inf
            Xdup = np.vstack([Xq[:8], Xq[:8]])  # duplicated points, no noise, no jitter: an exactly singular covariance
            z = np.zeros((1, 16))
            while not stop.is_set():
                try:
                    ctx.sample_y(0, hk, Xdup, z, jitter=0.0)
                except lib.NotPositiveDefinite:
                    counts["notpd"] += 1
            ctx.close()
        except Exception as exc:  # pragma: no cover
            errs.append(exc)

    def working():
        try:
            ctx = lib.Context(X, y, 1e-10, max_batch=4)  # 12 proposals in chunks of 4: downloads through the arena
            ctx.posterior(H[:1])
            m0, v0 = ctx.predict(H[:1], Xq)
            for _ in range(150):
                got = np.full(len(H), np.nan)
                got[:] = ctx.lml(H)
                np.testing.assert_allclose(got, ref, rtol=RTOL)
                m1, v1 = ctx.predict(H[:1], Xq)
                np.testing.assert_array_equal(m1, m0)
                np.testing.assert_array_equal(v1, v0)
                counts["ok"] += 1
            ctx.close()
        except Exception as exc:
            errs.append(exc)
        finally:
            stop.set()

    ta, tb = threading.Thread(target=failing), threading.Thread(target=working)
    ta.start()
    tb.start()
    tb.join()
    ta.join()
    assert not errs, errs
    assert counts["ok"] == 150 and counts["notpd"] > 0, counts


@pytest.mark.parametrize("n,d,B,stationary,form,warp", [
    (2048, 16, 128, "matern52", "product", False),  # BASELINE config C's batch: four-panel groups, two walker-group streams
    (1000, 8, 64, "matern52", "product", False),    # 8 block columns (two-panel groups), ragged last block
    (1290, 16, 32, "matern32", "sum", False),       # all 16 staged dimensions real; 11 block columns, 10 ragged rows
    (1290, 32, 24, "matern32", "sum", False),       # more than 16 input dimensions: not generated (nothing to differ)
    (700, 3, 104, "rbf", "product", False),         # 3 of 16 staged dimensions are real
    (515, 5, 288, "matern12", "sum", False),        # three rows in the last block column; two chunks of max_batch = 144
    (900, 4, 72, "matern52", "product", True),      # per-walker warped inputs
])
def test_gram_blocks_generated_inside_the_trailing_update_are_the_gram_kernels(lib, O, monkeypatch, n, d, B, stationary, form,
                                                                               warp):
    """The first panel group's updates generate the kernel-matrix blocks they touch first in their accumulators (S4GenF,
    csrc/bgp_s4.h) and the Gram kernel builds block column 0 only; BGP_SYRK_GEN=0 builds every block with the Gram kernel as
    before.  Same arithmetic per element: the log-likelihoods of the two schedules are bit-identical, and they are the oracle's."""
    X, y = synth(n, d, 7 + n)
    rs = np.random.RandomState(n + B)
    H = np.concatenate([[0.2], np.full(d, np.log(0.5)), [np.log(0.03)]]) + 0.2 * rs.randn(B, d + 2)
    H[1, -1] = -np.inf  # (noise level 0: the White term drops out of the diagonal)
    W = 0.3 * rs.randn(B, 2 * d) if warp else None
    ctx = lib.Context(X, y, 1e-10, form=form, stationary=stationary, max_batch=min(B, 144))
    ctx.set_persist(0)  # the launch schedule (the launch-free kernel has its own generation)
    if n != 2048:
        ctx.set_streams(1)  # (config C's batch keeps the automatic two walker-group streams: 64 matrices each; the smaller batches
        # would fall below the rule's size per group)
    monkeypatch.setenv("BGP_SYRK_GEN", "0")
    ref = ctx.lml_warped(H, W) if warp else ctx.lml(H)
    assert ctx.gen_stats() == {"batches": 0, "launches": 0}
    monkeypatch.setenv("BGP_SYRK_GEN", "1")
    got = ctx.lml_warped(H, W) if warp else ctx.lml(H)
    stats = ctx.gen_stats()
    ctx.close()
    assert np.array_equal(got, ref), float(np.max(np.abs(got - ref)))
    if d <= 16:
        # every chunk / walker group of the call went that way: the look-ahead columns of the first panel group (P - 1 of them,
        # P = 4 from 12 block columns, else 2) and its bulk update generate
        nblk = -(-n // 128)
        per_batch = min(4 if nblk >= 12 else 2, nblk)
        assert stats["batches"] >= 1 and stats["launches"] == per_batch * stats["batches"], stats
    else:
        assert stats == {"batches": 0, "launches": 0}
    if not warp and n <= 1300:
        rows = [0, 2, 3, 4, 5]  # (row 1 has no noise: conditioned by the 1e-10 jitter alone, it is no fixed point for a tolerance)
        np.testing.assert_allclose(got[rows], O.lml_batch(X, y, np.full(n, 1e-10), H[rows], stationary, form), rtol=RTOL)
