"""N>1 path on CPU: world_size-2 gloo job (torch.distributed.run), independent sub-ensembles per rank,
final all-gather of the chains, max-over-ranks timing."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gloo_chain_gather(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"),
           str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    r = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    for k in range(2):
        assert r[k]["ws"] == 2 and r[k]["rank"] == k
        assert r[k]["local_shape"] == [20 * 12, 3]
        assert r[k]["all_shape"] == [2 * 20 * 12, 3]  # rank-major concatenation on every rank
        assert r[k]["own_slice_ok"]
        assert r[k]["tmax"] == 2.0
    assert r[0]["checksum_all"] == r[1]["checksum_all"]
    assert np.isclose(r[0]["checksum_local"] + r[1]["checksum_local"], r[0]["checksum_all"])
    assert r[0]["checksum_local"] != r[1]["checksum_local"]  # independent sub-ensembles (rank seeds differ)


def test_single_process_helpers():
    import bayes_skopt_amd as bask

    d = bask.distributed
    assert d.world() == (0, 0, 1) or d.world()[2] >= 1
    c = np.arange(12.0).reshape(4, 3)
    np.testing.assert_array_equal(d.gather_chains(c), c)
    assert d.max_over_ranks(3.5) == 3.5
    assert d.rank_seed(0, 0) != d.rank_seed(0, 1)
