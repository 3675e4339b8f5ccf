#!/usr/bin/env python3
"""WBPR (popularity-weighted negatives, recom_wbpr.pyx:131-139) at the ML-20M shape: the LDS-bin form against the fused
atomic kernel — kernel ms per epoch (HIP events), 'correct' fraction and skip rate of the last epoch, pairwise accuracy
on a fixed probe sample whose negatives are popularity-drawn."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cornac_amd import _lib  # noqa: E402

k, epochs, lr, reg = 64, 8, 0.05, 0.01
n_users, n_items, indptr, indices = bench.load_dataset("ml20m", 0, os.environ.get("TMPDIR", "/tmp"))
nnz = len(indices)
user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
rs = np.random.RandomState(1)
pick = rs.randint(nnz, size=400000)
pu, pi, pj = user_ids[pick], indices[pick], indices[rs.randint(nnz, size=400000)]
for name, flags in (("fused", _lib.FORM_FUSED), ("ldsbin", _lib.FORM_LDSBIN)):
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    U, V, B = bench.init_factors(n_users, n_items, k, 100)
    tr.set_factors(U, V, B)
    tr.seed_hogwild(0xC0FFEE)
    tr.fit_epochs(1, lr, reg, True, _lib.NEG_POPULARITY, _lib.MODE_HOGWILD, flags)
    tr.kernel_timing(True)
    c, s = tr.fit_epochs(epochs - 1, lr, reg, True, _lib.NEG_POPULARITY, _lib.MODE_HOGWILD, flags)
    ms, launches = tr.kernel_timing(False)
    c, s = tr.fit_epochs(1, lr, reg, True, _lib.NEG_POPULARITY, _lib.MODE_HOGWILD, flags)
    U2, V2, B2 = tr.get_factors()
    tr.close()
    x = B2[pi] - B2[pj] + np.einsum("nk,nk->n", U2[pu], V2[pi] - V2[pj])
    print("%-7s %.3f ms/epoch (%d launches) = %.3f G triplets/s | correct %.4f skipped %.4f | probe accuracy %.4f loss %.4f"
          % (name, ms / (epochs - 1), launches, nnz * (epochs - 1) / ms / 1e6, c / max(nnz - s, 1), s / nnz, float(np.mean(x > 0)),
             float(np.mean(np.log1p(np.exp(-x))))), flush=True)
