#!/usr/bin/env python3
"""Deterministic MF epoch at the Netflix Prize shape (480 189 x 17 770, 100 480 507 ratings, k = 128): time of the dataflow
kernel (one persistent launch) for a few stored orders — the synthetic set as generated (sorted by user), its 1 024-block
shuffle (the GPU test's order), sorted by item (the real Netflix files' order) and a full random permutation.
--every N keeps every N-th rating.  --check compares the first order's result with the sequential oracle."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_ratings  # noqa: E402
from cornac_amd import _lib, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--every", type=int, default=1)
ap.add_argument("--orders", default="blocks,by_item,random")
ap.add_argument("--check", action="store_true")
args = ap.parse_args()
n_users, n_items, nnz, zipf_a, seed = synth.CONFIGS["netflix"]
rid, cid, val = synth_ratings(n_users, n_items, nnz, zipf_a, seed)
k, lr, reg = 128, 0.01, 0.02
rs = np.random.RandomState(11)
U0 = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
V0 = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
zu, zi = np.zeros(n_users, np.float32), np.zeros(n_items, np.float32)
mu = float(np.float32(val.mean(dtype=np.float64)))
for name in args.orders.split(","):
    if name == "by_user":
        order = np.arange(nnz)
    elif name == "blocks":
        r2 = np.random.RandomState(11)
        blocks = r2.permutation(1024)
        cuts = np.linspace(0, nnz, 1025).astype(np.int64)
        order = np.concatenate([np.arange(cuts[b], cuts[b + 1]) for b in blocks])
    elif name == "by_item":
        order = np.argsort(cid, kind="stable")
    else:
        order = np.random.RandomState(5).permutation(nnz)
    order = order[::args.every]
    r_, c_, v_ = (np.ascontiguousarray(x[order]) for x in (rid, cid, val))
    tr = _lib.MfTrainer(r_, c_, v_, n_users, n_items, k)
    tr.set_factors(U0, V0, zu, zi)
    t0 = time.perf_counter()
    loss, _ = tr.fit(1, lr, reg, mu, True, False, _lib.MODE_DETERMINISTIC)
    dt = time.perf_counter() - t0
    timing = tr.last_timing()
    t0 = time.perf_counter()
    tr.fit(1, lr, reg, mu, True, False, _lib.MODE_DETERMINISTIC)
    dt2 = time.perf_counter() - t0
    Ud, Vd, Bud, Bid = tr.get_factors()
    tr.close()
    line = "%-8s %10d ratings: first fit %.2f s (schedule %.2f s, kernel %.2f s), second epoch %.2f s = %.1f M ratings/s" % (
        name, len(v_), dt, timing["schedule_ms"] / 1e3, timing["sgd_ms"] / 1e3, dt2, len(v_) / dt2 / 1e6)
    if args.check:
        from oracle import oracle

        Uo, Vo, Buo, Bio = U0.copy(), V0.copy(), zu.copy(), zi.copy()
        lo = np.zeros(2, np.float32)
        t0 = time.perf_counter()
        oracle.lib().oracle_mf_fit(r_, c_, v_, len(v_), Uo, Vo, Buo, Bio, k, lr, reg, mu, 2, 1, 1, 0, lo.ctypes.data)
        t_cpu = (time.perf_counter() - t0) / 2
        err = max(np.abs(Ud - Uo).max(), np.abs(Vd - Vo).max(), np.abs(Bud - Buo).max(), np.abs(Bid - Bio).max())
        line += " | max |err| vs the sequential oracle after 2 epochs %.3g; oracle (1 thread) %.1f s per epoch" % (err, t_cpu)
    print(line, flush=True)
