#!/usr/bin/env python3
"""In-kernel timeline of ONE launch-free factorisation in its default form (the chain workgroup solves block (J+1, J) and
updates block (J+1, J+1) itself): what a block column costs the chain -- pf_block, the wait for the tile workers'
pre-updates, its own solve + update -- and what the tile tasks spend their time on.  usage: persist_trace.py n d B"""
import os
import sys

os.environ["BGP_PERSIST"] = "1"
os.environ["BGP_PS_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bayes_skopt_amd  # noqa: E402,F401
from bayes_skopt_amd import _lib  # noqa: E402

n, d, B = (int(a) for a in sys.argv[1:4])
rng = np.random.RandomState(0)
X = rng.uniform(size=(n, d))
y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.randn(n)
y = (y - y.mean()) / y.std()
ctx = _lib.Context(X, y, np.full(n, 1e-10), max_batch=B)
H = np.concatenate([[0.0], np.full(d, np.log(0.3)), [np.log(0.01)]]) + 0.05 * rng.randn(B, d + 2)
for _ in range(4):
    ctx.lml(H)
tr = ctx.ps_trace()
assert tr is not None, "no trace: was BGP_PS_TRACE=1 in place before the context was created?"
ch, tl = tr[0].astype(np.int64), tr[1].astype(np.int64)
Bt, nblk, total = ch.shape[0], ch.shape[1], tl.shape[0]
t0 = ch[:, 0, 0].min()
us = lambda v: (v - t0) / 100.0
print(f"n={n} B={B} nblk={nblk} tile tasks={total}; times in us")
print("chain, matrix 0:  J | pf_block start | pf_block | wait for pre-updates | own solve + update | column")
for J in range(nblk):
    a = ch[0, J]
    nxt = ch[0, J + 1, 0] if J + 1 < nblk else a[1]
    last = J + 1 == nblk
    print(f"  {J:2d} | {us(a[0]):9.1f} | {(a[1]-a[0])/100:6.1f} | " + ("" if last else f"{(a[2]-a[1])/100:6.1f} | {(a[3]-a[2])/100:6.1f} | {(nxt-a[0])/100:6.1f}"))
for name, i0, i1 in (("flags -> solve issued", 2, 4), ("epilogue + X stored and drained", 4, 5), ("X -> LDS", 5, 6),
                     ("diagonal update", 6, 7), ("tile -> LDS", 7, 3)):
    v = (ch[:, :-1, i1] - ch[:, :-1, i0]) / 100.0
    print(f"  own share: {name:34s} mean {v.mean():6.1f}  max {v.max():6.1f}")
pf = (ch[:, :, 1] - ch[:, :, 0]) / 100.0
wt = (ch[:, :-1, 2] - ch[:, :-1, 1]) / 100.0
cr = (ch[:, :-1, 3] - ch[:, :-1, 2]) / 100.0
print("means over matrices: pf_block %.1f us, wait %.1f (columns > 0: %.1f, max %.1f), own share %.1f; chains end at %.1f .. %.1f us" % (
    pf.mean(), wt.mean(), wt[:, 1:].mean() if nblk > 2 else 0.0, wt.max(), cr.mean(), us(ch[:, -1, 1].min()), us(ch[:, -1, 1].max())))
if total:
    done = tl[:, 6] > 0
    print("tile tasks done:", int(done.sum()), "of", total, " last finishes at %.1f us" % us(tl[:, 6].max()))
    meta = tl[:, 7]
    kind = (meta >> 28) & 0xf   # 0: panel solve S(I, J), 1: pre-update of block (I, I-1), 2: of block (I, I)
    Jc = (meta >> 20) & 0xff
    I = (meta >> 12) & 0xff
    bm = meta & 0xfff

    def stat(name, v):
        v = v[np.isfinite(v)]
        if len(v):
            print(f"  {name:52s} mean {v.mean():7.1f}  median {np.median(v):7.1f}  p90 {np.percentile(v, 90):7.1f}  max {v.max():7.1f}")

    d01 = np.where(tl[:, 1] > 0, (tl[:, 1] - tl[:, 0]) / 100.0, np.nan)
    d23 = np.where(tl[:, 2] > 0, (tl[:, 3] - tl[:, 2]) / 100.0, np.nan)
    for nm, sel in (("S d=2", done & (kind == 0) & (I - Jc == 2)), ("S d>=3", done & (kind == 0) & (I - Jc >= 3)),
                    ("P", done & (kind == 1)), ("Dg", done & (kind == 2))):
        if not sel.any():
            continue
        stat(f"{nm}: ticket -> first panels ready", d01[sel])
        stat(f"{nm}: last panel ready -> block stored", d23[sel])
        if nm.startswith("S"):
            stat(f"{nm}: stored (or ticket) -> W ready", ((tl[:, 4] - tl[:, 3]) / 100.0)[sel])
            stat(f"{nm}: W ready -> solved", ((tl[:, 5] - tl[:, 4]) / 100.0)[sel])
        stat(f"{nm}: computed -> drained, released, flag set", ((tl[:, 6] - (tl[:, 5] if nm.startswith("S") else tl[:, 3])) / 100.0)[sel])
        stat(f"{nm}: whole task", ((tl[:, 6] - tl[:, 0]) / 100.0)[sel])
    # slack of the two hand-overs to the chain, matrix 0: published this long before pf_block(I-1) ended (the chain asks then)
    for nm, k in (("P", 1), ("Dg", 2)):
        sel = np.flatnonzero(done & (kind == k) & (bm == 0))
        if len(sel):
            sel = sel[np.argsort(I[sel])]
            slack = (ch[0, I[sel] - 1, 1] - tl[sel, 6]) / 100.0
            print(f"  {nm}(I) of matrix 0 published this long before the chain asked (us, I = {I[sel][0]} ..):", " ".join(f"{v:.0f}" for v in slack))
    # how much of the time the tile workgroups hold a task is arithmetic: the factorisation's n^3 / 3 flop per matrix at one compute
    # unit's fp64 MFMA rate (0.307 TF) against the sum of the tasks' durations (ticket -> published) and against workers x call
    held = float(((tl[:, 6] - tl[:, 0])[done]).sum()) / 100.0
    floor_us = B * (n ** 3 / 3.0) / 0.307e12 * 1e6
    span = us(max(tl[:, 6].max(), ch[:, -1, 1].max()))
    print("  tile side: tasks hold workgroups for %.0f CU-us in all; the factorisations' flops are %.0f CU-us at the single-CU MFMA rate "
          "(%.0f %% of the held time); call span %.0f us" % (held, floor_us, 100.0 * floor_us / held, span))
