#!/usr/bin/env python3
"""rank() leg alone (for profiling): fused scoring GEMM + top-k over all users of an ML-20M-shaped model.
    python tools/bench_rank.py [--users 138493 --items 26744 --k 64 --topk 10 --repeats 5]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=138493)
ap.add_argument("--items", type=int, default=26744)
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--topk", type=int, default=10)
ap.add_argument("--repeats", type=int, default=5)
args = ap.parse_args()
rs = np.random.RandomState(0)
U = rs.normal(0, 0.3, (args.users, args.k)).astype(np.float32)
V = rs.normal(0, 0.3, (args.items, args.k)).astype(np.float32)
sc = _lib.Scorer(U, V, rs.normal(0, 0.1, args.items).astype(np.float32), None)
sc.rank_topk_device_ms(0, args.users, args.topk, 1)
ms = sc.rank_topk_device_ms(0, args.users, args.topk, args.repeats) / args.repeats   # the call returns the total
fl = 2.0 * args.users * args.items * args.k
print(json.dumps({"ms": ms, "tflops": fl / ms / 1e9, "mfma_frac": fl / ms / 1e9 / 157.3}))
