#!/usr/bin/env python3
"""What a half-step of the exact single-ensemble sharding costs ONE rank at N = 8 besides its device call: the sampler's
bookkeeping for all 128 rows, the log-priors of all rows, the submit / collect of its 16 rows.  One process, one GPU: the
group is faked (every rank's values = this rank's), so the chain is meaningless and the TIMING is that of rank 0 of 8 with
a free collective.  usage: shard_host_probe.py [ws]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import bayes_skopt_amd as bask  # noqa: E402
from bayes_skopt_amd import distributed  # noqa: E402
from bayes_skopt_amd.bayesgpr import _ShardedLogProb, _AsyncLogProb  # noqa: E402

ws = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gp, X, y, priors, theta0, pos = bench.setup_config_c(bask, 0, 0)
W, d = 256, 16
for shared in (False, True):
    if shared:
        os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", str(ws)
        distributed._state.update(backend="fake", rank=0, world=ws)
        distributed._allgather = lambda a: np.tile(np.asarray(a, dtype=np.float64), (ws,) + (1,) * np.ndim(a))
    lp = _ShardedLogProb(gp) if shared else _AsyncLogProb(gp)
    s = bask.sampler.EnsembleSampler(W, d + 2, lp, kwargs=dict(priors=priors))
    s.random_state = np.random.RandomState(1).get_state()
    st = s.run_mcmc(pos, 3)
    t0 = time.perf_counter()
    steps = 40
    st = s.run_mcmc(st.coords, steps, log_prob0=st.log_prob, skip_initial_state_check=True)
    dt = (time.perf_counter() - t0) / steps * 1e3
    print(("rank 0 of %d (16 rows on the device, collective free)" % ws) if shared else "one GPU, 128 rows", "ms per step %.3f" % dt, flush=True)
H = gp._canonical(pos[:128 // ws])
for _ in range(3):
    gp._ctx.lml(H)
t0 = time.perf_counter()
for _ in range(30):
    gp._ctx.lml(H)
print("device call of %d rows alone: %.3f ms" % (128 // ws, (time.perf_counter() - t0) / 30 * 1e3))
