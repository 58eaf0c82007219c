#!/usr/bin/env python3
"""In-kernel timeline of ONE launch-free factorisation with chain PAIRS (BGP_PS_PAIR=1): per block column of matrix 0 --
pf_block of the factorising workgroup, and what the other workgroup of the pair (the helper: the solve of block (J+1, J) and
the update of block (J+1, J+1) under that pf_block) was doing relative to it.  usage: pair_trace.py n d B"""
import os
import sys

os.environ["BGP_PERSIST"] = "1"
os.environ["BGP_PS_PAIR"] = "1"
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
ch = ctx.ps_trace()[0].astype(np.int64)
nblk = ch.shape[1]
t0 = ch[:, 0, 0].min()
print(f"n={n} B={B} nblk={nblk}; us.  helper columns are relative to the END of the partner's pf_block(J)")
print(" J | pf start | pf_block | helper: ready to start (rel. pf START) | rows 0-6 done | last row + z seen | steps done | X out | tile in LDS | period")
for m in (0, B - 1):
    print("matrix", m)
    for J in range(nblk):
        a = ch[m, J]
        per = (ch[m, J + 1, 0] - a[0]) / 100.0 if J + 1 < nblk else float("nan")
        if J + 1 < nblk:
            rel = lambda i: (a[i] - a[1]) / 100.0
            print(f"{J:2d} | {(a[0]-t0)/100:8.1f} | {(a[1]-a[0])/100:8.1f} | {(a[2]-a[0])/100:8.1f} | {rel(5):8.1f} | {rel(7):8.1f} | {rel(4):8.1f} | {rel(6):8.1f} | {rel(3):8.1f} | {per:6.1f}")
        else:
            print(f"{J:2d} | {(a[0]-t0)/100:8.1f} | {(a[1]-a[0])/100:8.1f}")
pf = (ch[:, :, 1] - ch[:, :, 0]) / 100.0
per = (ch[:, 1:, 0] - ch[:, :-1, 0]) / 100.0
print("means: pf_block %.1f us, column period %.1f us, chains end at %.1f .. %.1f us" % (pf.mean(), per.mean(), (ch[:, -1, 1].min() - t0) / 100.0, (ch[:, -1, 1].max() - t0) / 100.0))

# ---- the hand-over on the tile side: X_{J+1,J} out (helper of column J) -> the P slices of block (J+2, J+1) -> helper of column J+1
# may start.  (Tile-task trace slots collide when B % 8 != 0: only the slices whose slot survived are listed.)
tl = ctx.ps_trace()[1].astype(np.int64)
meta = tl[:, 7]
kind = (meta >> 28) & 0xF
Jc = (meta >> 20) & 0xFF
Ii = (meta >> 12) & 0xFF
bb = meta & 0xFFF
xcc = (meta >> 32) & 7
print("\nP slices of block (J+2, J+1), matrix 0: us after X_{J+1,J} went out (helper of column J); 'next helper ready' = its wait is over")
print(" J | slice xcc: start  sees last panel  stored  signalled ... | next helper ready")
for J in range(max(0, nblk - 12), nblk - 2):
    ref = ch[0, J, 6]
    rows = [k for k in range(tl.shape[0]) if tl[k, 0] and kind[k] == 1 and bb[k] == 0 and Ii[k] == J + 2 and Jc[k] == J + 1]
    s = "  ".join(f"[x{xcc[k]} {(tl[k,0]-ref)/100:6.1f} {(tl[k,2]-ref)/100:5.1f} {(tl[k,3]-ref)/100:5.1f} {(tl[k,6]-ref)/100:5.1f}]" for k in rows)
    print(f"{J:2d} | {s} | {(ch[0, J + 1, 2] - ref) / 100:5.1f} | helper steps done {(ch[0, J, 4] - ref) / 100:5.1f}")

# ---- the critical solve S(J+2, J) and the quadrants Q of its block, relative to the START of pf_block(J)
print("\nS(J+2, J) and its Q quadrants, matrix 0: us after pf_block(J) started   (pf_block(J) ends at 'pf end')")
print(" J | pf end | Q: [streamed panel begins, both blocks whole and acquired, stored, signalled] x4 | S: start, quadrants seen, A fragments in, steps done, rhs done, signalled")
for J in range(max(1, nblk - 12), nblk - 2):
    ref = ch[0, J, 0]
    qs = [k for k in range(tl.shape[0]) if tl[k, 0] and kind[k] == 3 and bb[k] == 0 and Ii[k] == J + 2 and Jc[k] == J]
    ss = [k for k in range(tl.shape[0]) if tl[k, 0] and kind[k] == 0 and bb[k] == 0 and Ii[k] == J + 2 and Jc[k] == J]
    sq = "  ".join(f"[{(tl[k,1]-ref)/100:5.1f} {(tl[k,2]-ref)/100:5.1f} {(tl[k,3]-ref)/100:5.1f} {(tl[k,6]-ref)/100:5.1f}]" for k in qs)
    st = "  ".join(f"[{(tl[k,0]-ref)/100:6.1f} {(tl[k,2]-ref)/100:5.1f} {(tl[k,4]-ref)/100:5.1f} {(tl[k,3]-ref)/100:5.1f} {(tl[k,5]-ref)/100:5.1f} {(tl[k,6]-ref)/100:5.1f}]" for k in ss)
    print(f"{J:2d} | {(ch[0, J, 1] - ref) / 100:5.1f} | {sq} | {st}")
