#!/usr/bin/env python3
"""Per-bin durations of the LDS-bin BPR epoch (profile build: CORNAC_HIP_PROFILE=1) regressed on the bin's cold / hot draw
counts: what a hot draw costs relative to a cold one, and how much of the epoch is the tail of the heaviest bin.
    CORNAC_HIP_PROFILE=1 python tools/ldsbin_clocks.py [--x 100] [--sg 16] [--hc 32]"""
import argparse
import os
import sys
import tempfile

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--x", type=int, default=100)
ap.add_argument("--sg", type=int, default=16)
ap.add_argument("--hc", type=int, default=32)
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--epochs", type=int, default=6)
args = ap.parse_args()
path = os.path.join(tempfile.gettempdir(), "ldsbin_clocks_%d.txt" % os.getpid())
os.environ["CORNAC_HIP_LDSBIN_CLOCKS"] = path
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cornac_amd import _lib  # noqa: E402

n_users, n_items, indptr, indices = bench.load_dataset("ml20m", 0, os.environ.get("TMPDIR", "/tmp"))
tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, args.k)
tr.ldsbin_config(hot_x1000=args.x)
tr.ldsbin_deal_config(strata_groups=args.sg, hot_cost_x16=args.hc)
tr.set_factors(*bench.init_factors(n_users, n_items, args.k, 100))
tr.seed_hogwild(0xC0FFEE)
for _ in range(args.epochs):
    tr.fit_epochs(1, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, 0)
tr.close()
d = np.loadtxt(path)
os.unlink(path)
for e in np.unique(d[:, 0])[1:]:
    m = d[d[:, 0] == e]
    start, end, cold, hot = m[:, 2] / 100.0, m[:, 3] / 100.0, m[:, 4], m[:, 5]  # us
    dur = end - start
    A = np.stack([cold, hot, np.ones_like(cold)], 1)
    coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
    res = dur - A @ coef
    print("epoch %d: last end %.0f us | start spread %.0f us | dur min/mean/max %.0f/%.0f/%.0f us | fit: %.4f us/cold + %.4f us/hot "
          "(hot = %.2f cold) + %.0f us, residual std %.0f us | draws max/mean %.4f" %
          (e, end.max(), start.max() - start.min(), dur.min(), dur.mean(), dur.max(), coef[0], coef[1], coef[1] / coef[0],
           coef[2], res.std(), (cold + hot).max() / (cold + hot).mean()))
