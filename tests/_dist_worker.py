"""Worker for tests/test_cpu_distributed.py: one rank of a world_size-2 gloo job.  Each rank runs an
independent sub-ensemble (the N>1 layout of bench.py: no collective in the sampling loop), then the
chains are all-gathered and timing is max-reduced."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bayes_skopt_amd as bask  # noqa: E402
from bayes_skopt_amd import distributed  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, local_rank, ws = distributed.init_process_group(backend="gloo")
    p, W, steps = 3, 12, 30
    mu = np.array([1.0, -2.0, 0.5])

    def log_prob(Xb):  # toy Gaussian target standing in for the device LML on a CPU-only box
        return -0.5 * ((Xb - mu) ** 2).sum(axis=1)

    rng = np.random.RandomState(distributed.rank_seed(0, rank))
    sampler = bask.sampler.EnsembleSampler(W, p, log_prob)
    sampler.random_state = np.random.RandomState(distributed.rank_seed(1, rank)).get_state()
    sampler.run_mcmc(mu + 1e-2 * rng.randn(W, p), steps)
    local = sampler.get_chain(flat=True, discard=10)
    distributed.barrier()
    allc = distributed.gather_chains(local)
    tmax = distributed.max_over_ranks(float(rank + 1))
    json.dump({"rank": rank, "ws": ws, "local_shape": local.shape, "all_shape": allc.shape,
               "own_slice_ok": bool(np.array_equal(allc[rank * len(local):(rank + 1) * len(local)], local)),
               "checksum_all": float(allc.sum()), "checksum_local": float(local.sum()), "tmax": tmax},
              open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
