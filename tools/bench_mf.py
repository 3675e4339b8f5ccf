#!/usr/bin/env python3
"""MF throughput probe (not the headline bench): ratings/s of the hogwild MF kernel on a
dataset-shaped synthetic rating set.   python tools/bench_mf.py --config ml20m --k 64"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="ml20m")
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--epochs", type=int, default=5)
ap.add_argument("--shuffle", action="store_true", help="shuffle the COO order (default: sorted by user)")
args = ap.parse_args()
n_users, n_items, nnz, a, seed = synth.CONFIGS[args.config]
path = "/tmp/cornac_amd_mf_%s.npz" % args.config
if os.path.exists(path):
    z = np.load(path); users, items = z["u"], z["i"]
else:
    users, items = synth.zipf_interactions(n_users, n_items, nnz, a, seed)
    np.savez(path, u=users, i=items)
rs = np.random.RandomState(1)
bu, bi = rs.normal(0, 0.5, n_users), rs.normal(0, 0.5, n_items)
val = np.clip(np.rint(3.5 + bu[users] + bi[items] + rs.normal(0, 0.7, len(users))), 1, 5).astype(np.float32)
if args.shuffle:
    p = rs.permutation(len(users)); users, items, val = users[p], items[p], val[p]
k = args.k
tr = _lib.MfTrainer(users, items, val, n_users, n_items, k)
U = rs.normal(0, 0.01, (n_users, k)).astype(np.float32); V = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
tr.set_factors(U, V, np.zeros(n_users, np.float32), np.zeros(n_items, np.float32))
mu = float(val.mean())
tr.fit(1, 0.01, 0.02, mu, True, False, _lib.MODE_HOGWILD)  # warm-up (builds ownership tables)
tr.kernel_timing(True)
t0 = time.perf_counter()
loss, n = tr.fit(args.epochs, 0.01, 0.02, mu, True, False, _lib.MODE_HOGWILD)
dt = time.perf_counter() - t0
kms, launches = tr.kernel_timing(False)
b = 16 * k + 16 + 20
print(json.dumps({"config": args.config, "k": k, "nnz": len(val), "ratings_per_s": len(val) * args.epochs / dt,
                  "ms_per_epoch": 1e3 * dt / args.epochs, "kernel_ms": kms / max(launches, 1),
                  "algorithmic_bytes_per_rating": b,
                  "roofline_frac": len(val) * b / (kms / max(launches, 1) / 1e3) / 8e12,
                  "loss": [float(x) for x in loss]}))
