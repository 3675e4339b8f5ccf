#!/usr/bin/env python3
"""Lost-update census of the XCD-strata epoch (profile build, ablation bit 4 of the high byte = 16): every triplet adds 1 to
element 0 of its positive item row and 1 to element 1 of its negative row instead of the SGD deltas.  The same
sampler with every row atomic (hot_permille 1000, min_mult 0) gives the exact touch counts; the plain
read-modify-write run's deficit is the number of updates lost to the intra-XCD race, reported by popularity decile."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cornac_amd import _lib  # noqa: E402

assert _lib.PROFILE, "run with CORNAC_HIP_PROFILE=1"
if len(sys.argv) > 1 and sys.argv[1] == "big":
    # a table of 2^21 rows (262 144 per XCD partition), 2 M users x 5 interactions, k = 128: the regime where the form is
    # the automatic choice (>= 2^20 rows)
    n_users, n_items, d = 2_000_000, 1 << 21, 5
    rs = np.random.RandomState(45)
    base = rs.randint(0, n_items, size=n_users, dtype=np.int64)
    step = rs.randint(1, n_items // (2 * d), size=n_users, dtype=np.int64)
    it = (base[:, None] + step[:, None] * np.arange(d, dtype=np.int64)[None, :]) % n_items
    it.sort(axis=1)
    indices = it.astype(np.int32).ravel()
    indptr = (np.arange(n_users + 1, dtype=np.int64) * d).astype(np.int32)
    k = 128
else:
    n_users, n_items, indptr, indices = bench.load_dataset("ml20m", 0, os.environ.get("TMPDIR", "/tmp"))
    k = 64
flags = 16 << 8


def run(cfg, variant):
    os.environ["CORNAC_HIP_STRATA_VARIANT"] = str(variant)
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    tr.strata_config(**cfg)
    U, V, B = bench.init_factors(n_users, n_items, k, 100)
    tr.set_factors(U, np.zeros_like(V), B)
    tr.seed_hogwild(0xC0FFEE)
    tr.fit_epochs(1, 0.05, 0.0, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags | _lib.FORM_STRATA)
    _, V2, _ = tr.get_factors()
    tr.close()
    return V2[:, 0].astype(np.float64), V2[:, 1].astype(np.float64)


pos_x, neg_x = run(dict(hot_permille=1000, hot_min_mult_x100=0), 0)
deg = np.bincount(indices, minlength=n_items)
order = np.argsort(-deg, kind="stable")
print("exact touches: pos %d neg %d" % (pos_x.sum(), neg_x.sum()))
arms = (("h0 v0", dict(hot_permille=0), 0), ("h0 v1", dict(hot_permille=0), 1), ("h0 v3", dict(hot_permille=0), 3),
        ("h0 v4", dict(hot_permille=0), 4), ("h120 v0", dict(hot_permille=120), 0))
if k == 128:
    arms = (("default", dict(), 0), ("h0", dict(hot_permille=0), 0))
for arm, cfg, var in arms:
    pos, neg = run(cfg, var)
    line = "%-8s lost: pos %.4f neg %.4f | by popularity decile (pos / neg):" % (arm, 1 - pos.sum() / pos_x.sum(), 1 - neg.sum() / neg_x.sum())
    for d in range(10):
        sel = order[d * n_items // 10:(d + 1) * n_items // 10]
        line += " %.3f/%.3f" % (1 - pos[sel].sum() / max(pos_x[sel].sum(), 1), 1 - neg[sel].sum() / max(neg_x[sel].sum(), 1))
    top = order[:30]
    line += " | top-30 pos %.3f" % (1 - pos[top].sum() / max(pos_x[top].sum(), 1))
    print(line, flush=True)
