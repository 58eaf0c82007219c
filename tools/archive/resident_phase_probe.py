#!/usr/bin/env python3
"""Where a device-resident sampler run spends its wall time, seen from Python: bgp_mcmc_begin, the bgp_mcmc_steps calls (upload +
enqueue of a plan segment), bgp_mcmc_end (wait + download) and the Python in between (drawing the plan).  usage: n d W steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayes_skopt_amd as bask
from bayes_skopt_amd.bayesgpr import _AsyncLogProb
from bayes_skopt_amd import sampler as S
from sklearn.gaussian_process.kernels import WhiteKernel
n, d, W, steps = (int(a) for a in sys.argv[1:5])
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d)); y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n); y = (y - y.mean()) / y.std()
gp = bask.BayesGPR(kernel=bask.construct_default_kernel(list(range(d))), random_state=0, max_batch=W // 2)
gp.kernel_ = gp.kernel + WhiteKernel(noise_level=0.01); gp.noise_ = 0.01
gp.X_train_, gp.y_train_ = X, y; gp.y_train_mean_, gp.y_train_std_ = np.zeros(1), 1
gp._ensure_context(batch_hint=W // 2)
priors = bask.guess_priors(gp.kernel_)
theta0 = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]])
pos = theta0 + 1e-2 * gp.random_state.randn(W, d + 2)
smp = S.EnsembleSampler(W, d + 2, _AsyncLogProb(gp), kwargs=dict(priors=priors))
st = smp.run_mcmc(pos, 2)
# phases
ctx = gp._ctx
T = {"begin": 0.0, "steps": 0.0, "end": 0.0}
ob, os_, oe = ctx.mcmc_begin, ctx.mcmc_steps, ctx.mcmc_end
def wrap(name, f):
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] += time.perf_counter() - t0; return r
    return g
ctx.mcmc_begin, ctx.mcmc_steps, ctx.mcmc_end = wrap("begin", ob), wrap("steps", os_), wrap("end", oe)
t0 = time.perf_counter()
st = smp.run_mcmc(st.coords, steps, log_prob0=st.log_prob, skip_initial_state_check=True)
dt = time.perf_counter() - t0
print("total %.2f ms = %.1f us per half-step; begin %.2f ms, steps calls %.2f ms, end (wait + download) %.2f ms, python plan etc %.2f ms" % (
    dt * 1e3, dt / (2 * steps) * 1e6, T["begin"] * 1e3, T["steps"] * 1e3, T["end"] * 1e3, (dt - sum(T.values())) * 1e3))
