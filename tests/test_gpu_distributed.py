"""Exact single-ensemble sharding on the device (SURVEY.md 8(e) option 1): two ranks (gloo rendezvous,
both on GPU 0 -- the GPU box has one device) each evaluate half of every proposal block with the HIP
path and all-gather the log-probabilities; the chain must equal the single-process device chain."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_sharded_fit_equals_single_process(tmp_path):
    import bayes_skopt_amd as bask

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", BGP_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_shard_worker.py"),
           str(tmp_path), "gpu"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]

    rng = np.random.RandomState(0)
    X = rng.uniform(size=(96, 2))
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(96)
    gp = bask.BayesGPR(kernel=bask.construct_default_kernel([0, 1]), random_state=3, device=0, normalize_y=True)
    gp.fit(X, y, n_desired_samples=60, n_burnin=4, n_walkers_per_thread=20, progress=False)
    c0, c1 = np.load(tmp_path / "chain0.npy"), np.load(tmp_path / "chain1.npy")
    np.testing.assert_array_equal(c0, c1)          # all ranks hold the same ensemble
    u0 = np.load(tmp_path / "chain_unsharded0.npy")
    np.testing.assert_array_equal(c0, u0)          # ... and it is the unsharded chain, bit for bit,
    np.testing.assert_array_equal(c0, gp.chain_)   # in the worker and in this process (reproducible device path)
    r0 = json.load(open(tmp_path / "shard0.json"))
    assert np.isclose(r0["lp_sum"], gp.log_marginal_likelihood_value_, rtol=1e-12)
