#!/usr/bin/env python3
"""Gram blocks generated inside the launch-free kernel (BGP_PS_GEN=1) against a Gram kernel in front of it (BGP_PS_GEN=0) and
against the launch schedule: bit-identity of the log-likelihoods and wall ms per LML call.  usage: gen_probe.py [n,d,B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from persist_probe import run  # noqa: E402

if __name__ == "__main__":
    shapes = [(1024, 8, 32), (1024, 8, 16), (1024, 8, 8), (1024, 8, 24), (768, 8, 32), (1536, 8, 16), (2048, 16, 16), (2048, 16, 8),
              (3072, 16, 8), (512, 8, 32), (640, 4, 16)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    ref = run({"BGP_PERSIST": "0"}, shapes, 30)
    g0 = run({"BGP_PERSIST": "1", "BGP_PS_PAIR": "0", "BGP_PS_GEN": "0"}, shapes, 30)
    g1 = run({"BGP_PERSIST": "1", "BGP_PS_PAIR": "0", "BGP_PS_GEN": "1"}, shapes, 30)
    for k in ref:
        same = ref[k]["lml"] == g0[k]["lml"] == g1[k]["lml"] and ref[k]["status"] == g0[k]["status"] == g1[k]["status"]
        print(f"{k:14s} launches {ref[k]['ms']:7.3f}  launch-free {g0[k]['ms']:7.3f}  + gen inside {g1[k]['ms']:7.3f} ms   "
              f"bits {'same' if same else 'DIFFER'}  stable {g1[k]['stable']}")
