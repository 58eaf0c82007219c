#!/usr/bin/env python3
"""One device-resident sampler run (bgp_mcmc_begin_ex / _steps / _end) of a BASELINE shape for a kernel trace:
resident_probe.py n d W steps [sharded] -- `sharded`: the ensemble sharded over the process group (a one-rank RCCL group with
BGP_DIST_FORCE=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=... in the environment): the pack kernel and the RCCL
all-gather sit between the LML batch and the next step kernel on the context's stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayes_skopt_amd as bask  # noqa: E402
from bayes_skopt_amd import distributed  # noqa: E402
from bayes_skopt_amd.bayesgpr import _AsyncLogProb, _ShardedLogProb  # noqa: E402
from sklearn.gaussian_process.kernels import WhiteKernel  # noqa: E402

n, d, W, steps = (int(a) for a in sys.argv[1:5])
sharded = len(sys.argv) > 5 and sys.argv[5] == "sharded"
if sharded:
    distributed.init_process_group()
    assert distributed.backend() == "rccl", distributed.backend()
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d))
y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
y = (y - y.mean()) / y.std()
gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0, max_batch=W // 2)
gp.kernel_ = gp.kernel + WhiteKernel(noise_level=0.01)
gp.noise_ = 0.01
gp.X_train_, gp.y_train_ = X, y
gp.y_train_mean_, gp.y_train_std_ = np.zeros(1), 1
gp._ensure_context(batch_hint=W // 2)
priors = bask.guess_priors(gp.kernel_)
theta0 = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
pos = theta0 + 1e-2 * gp.random_state.randn(W, d + 2)
smp = bask.sampler.EnsembleSampler(W, d + 2, (_ShardedLogProb if sharded else _AsyncLogProb)(gp), kwargs=dict(priors=priors))
st = smp.run_mcmc(pos, 2)
t0 = time.perf_counter()
st = smp.run_mcmc(st.coords, steps, log_prob0=st.log_prob, skip_initial_state_check=True)
dt = time.perf_counter() - t0
print("%sresident runs %d, %.4f ms per half-step, %.0f evals/s" % ("sharded (one-rank RCCL group), " if sharded else "", smp.resident_runs,
                                                                   dt / (2 * steps) * 1e3, W * steps / dt))
if sharded:
    distributed.destroy_process_group()
