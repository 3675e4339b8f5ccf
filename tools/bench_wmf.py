#!/usr/bin/env python3
"""WMF step-throughput probe (not the headline bench): ms per Adam step (one batch of 128 items over ALL
users) and the MFMA rate of its three GEMM-shaped pieces (6 n_users B k flops per step) on a dataset-shaped
synthetic rating matrix.   python tools/bench_wmf.py --config netflix --k 128 --steps 50"""
import argparse, json, os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="ml20m")
ap.add_argument("--k", type=int, default=128)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--shape", default=None, help="users,items,nnz: uniform random matrix instead of a named config "
                "(the step cost depends on n_users, batch, k and the batch's nnz only)")
args = ap.parse_args()
if args.shape:
    n_users, n_items, nnz = (int(x) for x in args.shape.split(","))
    args.config = "uniform:" + args.shape
    rs0 = np.random.RandomState(0)
    keys = np.unique(rs0.randint(0, n_users * n_items, size=int(nnz * 1.05), dtype=np.int64))[:nnz]
    users, items = keys // n_items, keys % n_items
else:
    n_users, n_items, nnz, a, seed = synth.CONFIGS[args.config]
path = "/tmp/cornac_amd_mf_%s.npz" % args.config
if args.shape:
    pass
elif os.path.exists(path):
    z = np.load(path); users, items = z["u"], z["i"]
else:
    users, items = synth.zipf_interactions(n_users, n_items, nnz, a, seed)
    np.savez(path, u=users, i=items)
rs = np.random.RandomState(1)
val = rs.randint(1, 6, len(users)).astype(np.float32)
R = sp.csc_matrix((val, (users, items)), shape=(n_users, n_items))
k = args.k
tr = _lib.WmfTrainer(R, k)
lim = np.sqrt(6.0 / (n_users + k))
tr.set_factors(rs.uniform(-lim, lim, (n_users, k)).astype(np.float32), rs.uniform(-lim, lim, (n_items, k)).astype(np.float32))
perm = rs.permutation(n_items)
batches = [perm[s:s + args.batch] for s in range(0, n_items, args.batch)]
batches = [b for b in batches if len(b) == args.batch][: args.steps]
tr.fit_batches(batches[:3], 0.01, 0.01, 1.0, 0.01, 0.001)  # warm-up
tr.kernel_timing(True)
t0 = time.perf_counter()
loss = tr.fit_batches(batches, 0.01, 0.01, 1.0, 0.01, 0.001)
dt = time.perf_counter() - t0
dev_ms = tr.last_device_ms()
flops = 6.0 * n_users * args.batch * k * len(batches)
print(json.dumps({"config": args.config, "k": k, "n_users": n_users, "batch": args.batch, "steps": len(batches),
                  "ms_per_step_wall": 1e3 * dt / len(batches), "ms_per_step_device": dev_ms / len(batches),
                  "tflops": flops / (dev_ms / 1e3) / 1e12, "mfma_frac": flops / (dev_ms / 1e3) / 157.3e12,
                  "epoch_s_est": dev_ms / 1e3 / len(batches) * np.ceil(n_items / args.batch),
                  "loss_first_last": [float(loss[0]), float(loss[-1])]}))
