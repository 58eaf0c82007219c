#!/usr/bin/env python3
"""In-kernel timeline of ONE launch-free factorisation (BGP_PERSIST=1 BGP_PS_TRACE=1 are set here): where the diagonal-block
chain waits, what the tile tasks spend their time on.  usage: persist_trace.py n d B"""
import ctypes as C
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
lib = _lib.load()
lib.bgp_debug_ps_trace.restype = C.c_int
lib.bgp_debug_ps_trace.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.c_size_t]
dims = (C.c_int * 3)()
lib.bgp_debug_ps_trace(ctx._h, dims, None, 0)
Bt, nblk, total = dims[0], dims[1], dims[2]
buf = np.zeros(Bt * nblk * 4 + total * 8, dtype=np.uint64)
assert lib.bgp_debug_ps_trace(ctx._h, dims, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0
ch = buf[: Bt * nblk * 4].reshape(Bt, nblk, 4).astype(np.int64)
tl = buf[Bt * nblk * 4:].reshape(total, 8).astype(np.int64)
t0 = min(ch[:, 0, 0].min(), tl[:, 0][tl[:, 0] > 0].min())
us = lambda v: (v - t0) / 100.0
print(f"n={n} B={B} nblk={nblk} tasks={total}; all times in us from the first stamp")
print("chain (matrix 0):  J | wait begin | wait end (=potrf start) | factorised | published |  wait   potrf  publish")
for J in range(nblk):
    a = ch[0, J]
    print(f"  {J:2d} | {us(a[0]):9.1f} | {us(a[1]):9.1f} | {us(a[2]):9.1f} | {us(a[3]):9.1f} | {(a[1]-a[0])/100:6.1f} {(a[2]-a[1])/100:6.1f} {(a[3]-a[2])/100:6.1f}")
w = (ch[:, :, 1] - ch[:, :, 0]) / 100.0
p = (ch[:, :, 2] - ch[:, :, 1]) / 100.0
r = (ch[:, :, 3] - ch[:, :, 2]) / 100.0
print("chain means over matrices: wait %.1f us/col (cols>0: %.1f), potrf %.1f, publish %.1f; chain ends at %.1f us" % (
    w.mean(), w[:, 1:].mean(), p.mean(), r.mean(), us(ch[:, -1, 3].max())))
meta = tl[:, 7]
J = (meta >> 24) & 0xff
I = (meta >> 16) & 0xff
diag = I == J
done = tl[:, 6] > 0
print("tile tasks done:", int(done.sum()), "of", total, " last finishes at %.1f us" % us(tl[:, 6].max()))
def stat(name, v):
    v = v[np.isfinite(v)]
    if len(v):
        print(f"  {name:46s} mean {v.mean():7.1f}  median {np.median(v):7.1f}  p90 {np.percentile(v, 90):7.1f}  max {v.max():7.1f}")
od = done & ~diag
dg = done & diag
d01 = np.where(tl[:, 1] > 0, (tl[:, 1] - tl[:, 0]) / 100.0, np.nan)
d12 = np.where((tl[:, 2] > 0) & (tl[:, 1] > 0), (tl[:, 2] - tl[:, 1]) / 100.0, np.nan)
d23 = np.where(tl[:, 2] > 0, (tl[:, 3] - tl[:, 2]) / 100.0, np.nan)
stat("off-diag: ticket -> first panels ready", d01[od])
stat("off-diag: first ready -> last panel ready", d12[od])
stat("off-diag: last panel ready -> C stored", d23[od])
stat("off-diag: C stored (or ticket) -> W ready", ((tl[:, 4] - tl[:, 3]) / 100.0)[od])
stat("off-diag: W ready -> solved", ((tl[:, 5] - tl[:, 4]) / 100.0)[od])
stat("off-diag: solved -> published", ((tl[:, 6] - tl[:, 5]) / 100.0)[od])
stat("off-diag: whole task", ((tl[:, 6] - tl[:, 0]) / 100.0)[od])
stat("diag: last panel ready -> stored", d23[dg])
stat("diag: stored -> published", ((tl[:, 6] - tl[:, 3]) / 100.0)[dg])
# critical path of matrix 0: potrf(J) published -> X(J+1,J) published -> diag(J+1) published -> potrf(J+1) starts
b = meta & 0x7fff
print("critical path, matrix 0:  J | potrf published -> (J+1,J) W seen | -> solved | -> published | diag(J+1): last ready | -> stored | -> published | -> potrf(J+1) starts")
for Jc in range(nblk - 1):
    m1 = np.flatnonzero((b == 0) & (J == Jc) & (I == Jc + 1))
    m2 = np.flatnonzero((b == 0) & (J == Jc + 1) & (I == Jc + 1))
    if len(m1) < 1 or len(m2) < 1:
        continue
    pub = ch[0, Jc, 3]
    wseen = tl[m1, 4].max(); solved = tl[m1, 5].max(); xpub = tl[m1, 6].max()
    lr = tl[m2, 2].max(); st = tl[m2, 3].max(); dp = tl[m2, 6].max()
    print(f"  {Jc:2d} | {(wseen-pub)/100:6.1f} | {(solved-wseen)/100:6.1f} | {(xpub-solved)/100:6.1f} | {(lr-xpub)/100:6.1f} | {(st-lr)/100:6.1f} | {(dp-st)/100:6.1f} | {(ch[0, Jc+1, 1]-dp)/100:6.1f}   total {(ch[0, Jc+1, 1]-pub)/100:6.1f}")
