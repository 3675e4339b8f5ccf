#!/usr/bin/env python3
"""Throughput probe of the row-sharded item table path (multi-GPU regime 2).  Launch under torchrun for N > 1
(one rank per GPU); with one rank the all-to-all still goes through RCCL.
    python tools/bench_sharded.py --users 2000000 --items 1000000 --degree 20 --k 128 --micro-batch 2000000"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from cornac_amd import _lib
from cornac_amd.dist import RowShardedBprTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=500_000)
ap.add_argument("--degree", type=int, default=20)
ap.add_argument("--k", type=int, default=128)
ap.add_argument("--micro-batch", type=int, default=2_000_000)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--trace", action="store_true", help="print the stage timeline (device events) of the last epoch")
args = ap.parse_args()
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
rs = np.random.RandomState(7 + rank)
nu, ni, d, k = args.users, args.items, args.degree, args.k
# fixed-degree synthetic rows (sorted, distinct): stride sampling over the item range
start = rs.randint(0, ni, nu).astype(np.int64)
step = ni // d
indices = ((start[:, None] + np.arange(d)[None, :] * step) % ni)
indices.sort(axis=1)
indptr = (np.arange(nu + 1, dtype=np.int64) * d).astype(np.int32)
tr = _lib.BprTrainer(indptr, indices.astype(np.int32).ravel(), nu, ni, nu, ni, k, device=local)
tr.set_factors(((rs.uniform(0, 1, (nu, k)).astype(np.float32) - 0.5) / k), None, None)
tr.seed_hogwild(11 + rank)
sh = RowShardedBprTrainer(tr, ni, k, dev, micro_batch=args.micro_batch)
V = ((np.random.RandomState(1).uniform(0, 1, (ni, k)).astype(np.float32) - 0.5) / k)
sh.load_items(V, np.zeros(ni, np.float32))
nnz = nu * d
sh.run(min(nnz, args.micro_batch), 0.05, 0.01); sh.finish()  # warm-up
dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter()
sh.rows_fetched = 0
sh._valid_draws.zero_()
for e in range(args.epochs):
    if args.trace and e == args.epochs - 1:
        sh.trace = []
    sh.run(nnz, 0.05, 0.01)
c, s = sh.finish()
dist.barrier(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
t = torch.tensor([dt], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    n = nnz * args.epochs * world
    print(json.dumps({"world": world, "users_per_rank": nu, "items": ni, "k": k, "nnz_per_rank": nnz,
                      "micro_batch": args.micro_batch, "triplets_per_s": n / t.item(),
                      "rows_fetched_per_triplet": sh.rows_fetched / max(sh.triplets, 1),
                      "exchange_bytes_per_triplet": sh.rows_fetched / max(sh.triplets, 1) * (k + 1) * 4 * 2}))
if args.trace and rank == 0:
    t0e = sh.trace[0][2]
    for label, which, ev in sh.trace:
        print("%-10s %s %9.3f ms" % (label, {"A": "", "B": "      ", "C": "            "}[which], t0e.elapsed_time(ev)))
tr.close()
dist.destroy_process_group()
