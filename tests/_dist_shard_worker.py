"""Worker for the exact single-ensemble sharding test (SURVEY.md 8(e) option 1): every rank drives the
same sampler RNG, evaluates only its rows of each proposal block and all-gathers the log-probabilities.
argv: out_dir mode [W]  (mode "toy": CPU toy target with W walkers (default 14); "fail": the same, rank 3's share raises in
its 6th call -- every rank must stop with ShardedEvaluationError; mode "gpu": the device LML, ranks share GPU 0; mode "gpu_generic":
the same with a kernel tree that has no canonical device form -- Matern() + Matern(): host-evaluated kernel matrices, device
factorisation -- whose finished values are gathered from the host)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bayes_skopt_amd as bask  # noqa: E402
from bayes_skopt_amd import distributed  # noqa: E402


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    rank, local_rank, ws = distributed.init_process_group(backend="gloo")
    calls = []
    if mode in ("toy", "fail"):
        import time

        p, W, steps = 3, (int(sys.argv[3]) if len(sys.argv) > 3 else 14), 25   # default W/2 = 7 rows per half-step: 3/4 over two ranks
        mu = np.array([1.0, -2.0, 0.5])

        def log_prob(Xb):
            calls.append(len(Xb))
            if mode == "fail" and rank == 3 and len(calls) == 6:
                json.dump({"t": time.time()}, open(os.path.join(out_dir, "raised.json"), "w"))
                raise FloatingPointError("rank 3's share of the block blew up")
            lp = -0.5 * ((Xb - mu) ** 2).sum(axis=1)
            lp[Xb[:, 0] > 1.5] = -np.inf  # hard wall: -inf must survive the gather
            return lp

        sampler = bask.sampler.EnsembleSampler(W, p, distributed.shard_log_prob(log_prob))
        sampler.random_state = np.random.RandomState(5).get_state()
        try:
            sampler.run_mcmc(mu + 1e-2 * np.random.RandomState(4).randn(W, p), steps)
        except distributed.ShardedEvaluationError as exc:
            json.dump({"t": time.time(), "msg": str(exc), "cause": repr(exc.__cause__), "calls": len(calls)},
                      open(os.path.join(out_dir, f"stopped{rank}.json"), "w"))
            raise
        chain = sampler.get_chain(flat=True)
        lp = sampler.get_log_prob(flat=True)
    else:
        rng = np.random.RandomState(0)
        X = rng.uniform(size=(96, 2))
        y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(96)
        def kernel():
            if mode == "gpu_generic":
                from sklearn.gaussian_process import kernels as sk

                return sk.Matern(length_scale=0.5, nu=2.5) + sk.Matern(length_scale=2.0, nu=1.5)
            return bask.construct_default_kernel([0, 1])

        gp = bask.BayesGPR(kernel=kernel(), random_state=3, device=0, normalize_y=True, shard_ensemble=True)
        gp.fit(X, y, n_desired_samples=60, n_burnin=4, n_walkers_per_thread=20, progress=False)
        assert gp._generic == (mode == "gpu_generic")
        chain = gp.chain_
        lp = np.array([gp.log_marginal_likelihood_value_])
        # the same fit without sharding, in this very process (same libraries / thread settings)
        gp1 = bask.BayesGPR(kernel=kernel(), random_state=3, device=0, normalize_y=True)
        gp1.fit(X, y, n_desired_samples=60, n_burnin=4, n_walkers_per_thread=20, progress=False)
        np.save(os.path.join(out_dir, f"chain_unsharded{rank}.npy"), gp1.chain_)
    np.save(os.path.join(out_dir, f"chain{rank}.npy"), chain)
    json.dump({"rank": rank, "ws": ws, "calls": calls, "lp_sum": float(np.sum(lp[np.isfinite(lp)]))},
              open(os.path.join(out_dir, f"shard{rank}.json"), "w"))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
