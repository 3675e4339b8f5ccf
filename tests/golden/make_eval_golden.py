#!/usr/bin/env python3
"""Generates tests/golden/eval_report.npz by RUNNING THE REAL REFERENCE (build container only): the reference's own
`ranking_eval` / `rating_eval` (cornac/eval_methods/base_method.py:35-226) over the reference's seeded BPR and MF on a
fixed train / test split.  tests/test_dropin_gpu.py fits cornac_amd's models on the same split ON THE DEVICE and
must reproduce the report.

    python tests/golden/make_eval_golden.py
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from oracle import build_ref, ref_loader  # noqa: E402

RANK_METRICS = [("Recall", 5), ("NDCG", 10), ("Precision", 3), ("AUC", None), ("MAP", None), ("MRR", None)]


def main():
    build_ref.build()
    ns = ref_loader.load()
    rm = ns.metrics
    rs = np.random.RandomState(17)
    nu, ni, n = 120, 90, 4200
    keys = rs.permutation(nu * ni)[:n]
    bu, bi = rs.normal(0, 0.7, nu), rs.normal(0, 0.7, ni)
    u, i = keys // ni, keys % ni
    r = np.clip(np.rint(3.0 + bu[u] + bi[i] + rs.normal(0, 0.6, n)), 1, 5)
    n_train = 3200
    rows = [(int(a), int(b), float(c)) for a, b, c in zip(u, i, r)]
    train = ns.Dataset.build(rows[:n_train])
    test = ns.Dataset.build(rows[n_train:], global_uid_map=train.uid_map, global_iid_map=train.iid_map, exclude_unknowns=True)
    kw = dict(k=8, max_iter=15, learning_rate=0.05, lambda_reg=0.01, seed=21)
    fx = {"users": u.astype(np.int64), "items": i.astype(np.int64), "ratings": r.astype(np.float64),
          "n_train": np.int64(n_train), "k": np.int64(kw["k"]), "epochs": np.int64(kw["max_iter"]),
          "lr": np.float64(kw["learning_rate"]), "reg": np.float64(kw["lambda_reg"]), "seed": np.int64(kw["seed"]),
          "rating_threshold": np.float64(3.0)}
    mk = lambda: [getattr(rm, nme)() if k is None else getattr(rm, nme)(k=k) for nme, k in RANK_METRICS]  # noqa: E731
    for tag, cls in (("bpr", ns.BPR), ("mf", ns.MF)):
        m = cls(**kw).fit(train)
        avg, per_user = ns.eval_methods.base_method.ranking_eval(m, mk(), train, test, rating_threshold=3.0)
        fx[tag + "_rank_avg"] = np.asarray(avg, np.float64)
        fx[tag + "_rank_users"] = np.asarray(sorted(per_user[0].keys()), np.int64)
        if tag == "mf":
            avg_r, _ = ns.eval_methods.base_method.rating_eval(m, [rm.RMSE(), rm.MAE()], test)
            fx["mf_rating_avg"] = np.asarray(avg_r, np.float64)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval_report.npz"), **fx)
    print({k: (v if v.ndim == 0 or v.size < 8 else v.shape) for k, v in fx.items() if "avg" in k})


if __name__ == "__main__":
    main()
