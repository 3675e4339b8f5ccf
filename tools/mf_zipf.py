#!/usr/bin/env python3
"""MF hogwild epoch at the Netflix Prize shape for two item skews — Zipf 0.45 (bench.py's: the most-rated title holds
0.23 % of the ratings, as in the real set) and SURVEY 8d's 0.8 (2.8 % on one row) — in both hogwild forms.
Prints ms per epoch (wall), ratings/s and the roofline fraction by the algorithmic 2084 bytes per rating."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_ratings  # noqa: E402
from cornac_amd import _lib, synth  # noqa: E402

n_users, n_items, nnz, _, seed = synth.CONFIGS["netflix"]
k, lr, reg = 128, 0.01, 0.02
for zipf in [float(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("0.45", "0.8"))]:
    users, items, val = synth_ratings(n_users, n_items, nnz, zipf, seed)
    top = np.bincount(items, minlength=n_items).max()
    rs = np.random.RandomState(1)
    U = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
    V = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
    mu = float(val.mean())
    for form, name in ((2, "block rotation"), (1, "fused atomic kernel"))[: (1 if len(sys.argv) > 2 else 2)]:
        tr = _lib.MfTrainer(users, items, val, n_users, n_items, k)
        tr.hogwild_form(form)
        tr.set_factors(U, V, np.zeros(n_users, np.float32), np.zeros(n_items, np.float32))
        tr.fit(1, lr, reg, mu, True, False, _lib.MODE_HOGWILD)
        t0 = time.perf_counter()
        loss, _ = tr.fit(3, lr, reg, mu, True, False, _lib.MODE_HOGWILD)
        dt = (time.perf_counter() - t0) / 3
        st = tr.hogwild_stats()
        tr.close()
        print("zipf %.2f (hottest item %.2f %% of the ratings) %-20s %.1f ms per epoch, %.2f G ratings/s, frac %.2f, mse %.4f %s"
              % (zipf, 100.0 * top / nnz, name, 1e3 * dt, nnz / dt / 1e9, nnz * 2084 / dt / 8e12, loss[-1] / nnz, st), flush=True)
