#!/usr/bin/env python3
"""MF at the Netflix Prize shape through dist.MfBlockRotationTrainer on ONE rank laid out like a node of `--virtual-world`
ranks (2 N item blocks, one cornac_hip_mf handle per block, the trained block copied on the communication stream as if it
travelled), next to the plain single-handle epoch: what the rotation's step granularity and re-binding cost on one device.
    python tools/bench_mf_rotation.py [--virtual-world 8 --epochs 4]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_ratings
from cornac_amd import _lib, synth
from cornac_amd.dist import MfBlockRotationTrainer, _DeviceMfBlockTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--virtual-world", type=int, default=8)
ap.add_argument("--epochs", type=int, default=4)
ap.add_argument("--zipf", type=float, default=0.45)
ap.add_argument("--form", type=int, default=0, help="1 fused atomic kernel, 2 the handle's block rotation, 0 the trainer's rule")
ap.add_argument("--rank-share", type=int, default=1, help="keep 1 / this of the users: the ratings ONE rank of a node holds")
args = ap.parse_args()
n_users, n_items, nnz, _, seed = synth.CONFIGS["netflix"]
k, lr, reg = 128, 0.01, 0.02
users, items, val = synth_ratings(n_users, n_items, nnz, args.zipf, seed)
if args.rank_share > 1:      # users dealt round-robin, as fit_mf_sharded deals them
    keep = users % args.rank_share == 0
    users, items, val = users[keep] // args.rank_share, items[keep], val[keep]
    n_users = (n_users + args.rank_share - 1) // args.rank_share
    nnz = len(val)
if args.form:
    _DeviceMfBlockTrainer.FORCE_FORM = args.form
mu = float(val.mean())
rs = np.random.RandomState(1)
U = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
V = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
tr = _lib.MfTrainer(users, items, val, n_users, n_items, k)
tr.set_factors(U, V, np.zeros(n_users, np.float32), np.zeros(n_items, np.float32))
tr.fit(1, lr, reg, mu, True, False, _lib.MODE_HOGWILD)
t0 = time.perf_counter()
loss_plain, _ = tr.fit(args.epochs, lr, reg, mu, True, False, _lib.MODE_HOGWILD)
plain = (time.perf_counter() - t0) / args.epochs
tr.close()
dev = torch.device("cuda", 0)
order = np.argsort(-np.bincount(items, minlength=n_items), kind="stable")
t0 = time.perf_counter()
rot = MfBlockRotationTrainer(users, items, val, n_users, n_items, k, dev, item_order=order, emulate_traffic=True,
                             virtual_world=args.virtual_world)
setup = time.perf_counter() - t0
rot.set_user_factors(U, np.zeros(n_users, np.float32))
rot.load_items(V, np.zeros(n_items, np.float32))
rot.run_epoch(lr, reg, mu)
rot.finish()
t0 = time.perf_counter()
sq = []
for _ in range(args.epochs):
    rot.run_epoch(lr, reg, mu)
    sq.append(rot.finish())
dt = (time.perf_counter() - t0) / args.epochs
rot.close()
print(json.dumps({"workload": "biased MF k=128, %d x %d, %d ratings, Zipf %.2f" % (n_users, n_items, nnz, args.zipf),
                  "plain_ms_per_epoch": 1e3 * plain, "rotation_ms_per_epoch": 1e3 * dt, "tax": dt / plain - 1.0,
                  "form": args.form, "rank_share": args.rank_share, "blocks": rot.nb, "rows_per_block": rot.W, "steps_per_epoch": rot.nb, "setup_s": setup,
                  "mse_plain_last": float(loss_plain[-1]) / nnz, "mse_rotation_last": 0.5 * sq[-1] / nnz,
                  "frac_plain": nnz * 2084 / plain / 8e12, "frac_rotation": nnz * 2084 / dt / 8e12}))
