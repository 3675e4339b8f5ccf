#!/usr/bin/env python3
"""Large-table probe toward BASELINE.json configs[4] (100 M users x 10 M items, k = 128): BPR hogwild on a
synthetic interaction set whose user table is far larger than the 256 MiB Infinity Cache, so the row
gathers / updates really come from HBM.  Every user gets `--degree` distinct items (Zipf-free: uniform
items), generated without a global dedupe so that 10^8-10^9 interactions are practical on the host.

    python tools/bench_scale.py --users 20000000 --items 2000000 --degree 5 --k 128
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=20_000_000)
ap.add_argument("--items", type=int, default=2_000_000)
ap.add_argument("--degree", type=int, default=5)
ap.add_argument("--k", type=int, default=128)
ap.add_argument("--epochs", type=int, default=2)
args = ap.parse_args()
nu, ni, d, k = args.users, args.items, args.degree, args.k
t0 = time.time()
rs = np.random.RandomState(45)
base = rs.randint(0, ni, size=nu, dtype=np.int64)
step = rs.randint(1, ni // (2 * d), size=nu, dtype=np.int64)
items = (base[:, None] + step[:, None] * np.arange(d, dtype=np.int64)[None, :]) % ni  # d distinct items per user
items.sort(axis=1)
assert (np.diff(items, axis=1) > 0).all()
indices = items.astype(np.int32).ravel()
indptr64 = np.arange(nu + 1, dtype=np.int64) * d
assert indptr64[-1] < 2 ** 31
indptr = indptr64.astype(np.int32)
del items, base, step
t_gen = time.time() - t0
t0 = time.time()
tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
t_create = time.time() - t0
# factors are initialised on the host in chunks to bound host memory
def init(n, seed):
    r = np.random.RandomState(seed)
    return ((r.uniform(0, 1, (n, k)).astype(np.float32) - 0.5) / k)
V = init(ni, 2)
if nu <= 30_000_000:
    U = init(nu, 1)
    tr.set_factors(U, V, np.zeros(ni, np.float32))
    del U
else:  # the library zero-fills its tables; a 50 GB host copy of U is not needed for a throughput probe
    tr.set_factors(None, V, np.zeros(ni, np.float32))
tr.seed_hogwild(7)
t0 = time.time()
tr.fit_epochs(1, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)  # warm-up: builds ownership tables
t_warm = time.time() - t0
tr.kernel_timing(True)
t0 = time.perf_counter()
c, s = tr.fit_epochs(args.epochs, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
dt = time.perf_counter() - t0
kms, launches = tr.kernel_timing(False)
nnz = len(indices)
probes = int(np.ceil(np.log2(d + 1)))
b_full = 24 * k + 16 + 8 + 8 + 4 * probes
rate = nnz * args.epochs / dt
print(json.dumps({"users": nu, "items": ni, "nnz": nnz, "k": k, "U_GB": nu * k * 4 / 1e9, "V_GB": ni * k * 4 / 1e9,
                  "host_gen_s": t_gen, "create_s": t_create, "first_epoch_incl_ownership_s": t_warm,
                  "triplets_per_s": rate, "ms_per_epoch": 1e3 * dt / args.epochs, "kernel_ms": kms / max(launches, 1),
                  "algorithmic_bytes_per_triplet": b_full, "roofline_frac": rate * b_full / 8e12,
                  "skipped_frac": s / (nnz * args.epochs), "correct_frac": c / max(nnz * args.epochs - s, 1)}))
