#!/usr/bin/env python3
"""Generates tests/golden/f64_small.npz and vebpr_f64.npz by RUNNING THE REAL REFERENCE with float64 init_params (build container only):
`_fit_sgd` is a fused-type function (cornac/models/bpr/recom_bpr.pyx:211-214), so float64 U / V / Bi train in double.
Stored: the inputs (interaction triplets in insertion order, the float64 initial tables) and what the compiled
reference's BPR and WBPR (seeded => one thread) learned, plus score() of three users.

    python tests/golden/make_f64_golden.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

from make_golden import dataset, synth_pairs  # noqa: E402
from oracle import build_ref, ref_loader  # noqa: E402


def main():
    build_ref.build()
    ns = ref_loader.load()
    nu, ni, nnz, k, epochs, lr, reg, seed = 90, 70, 1500, 12, 8, 0.05, 0.01, 17
    u, i, r = synth_pairs(nu, ni, nnz, 0.8, 9)
    ds = dataset(ns, u, i, r)
    rs = np.random.RandomState(4)
    init = {"U": (rs.rand(ds.num_users, k) - 0.5) / k, "V": (rs.rand(ds.num_items, k) - 0.5) / k,
            "Bi": 0.01 * rs.randn(ds.num_items)}
    fx = {"users": u, "items": i, "ratings": r, "k": np.int64(k), "epochs": np.int64(epochs), "lr": np.float64(lr),
          "reg": np.float64(reg), "seed": np.int64(seed), "init_U": init["U"], "init_V": init["V"], "init_Bi": init["Bi"],
          "score_users": np.array([0, nu // 2, ds.num_users - 1], np.int64)}
    for tag, cls in (("bpr", ns.BPR), ("wbpr", ns.WBPR)):
        m = cls(k=k, max_iter=epochs, learning_rate=lr, lambda_reg=reg, seed=seed,
                init_params={n: a.copy() for n, a in init.items()}).fit(ds)
        assert m.u_factors.dtype == np.float64
        fx[tag + "_U"], fx[tag + "_V"], fx[tag + "_B"] = m.u_factors.copy(), m.i_factors.copy(), m.i_biases.copy()
        fx[tag + "_scores"] = np.stack([m.score(int(x)) for x in fx["score_users"]])
        assert fx[tag + "_scores"].dtype == np.float64
    np.savez_compressed(os.path.join(HERE, "f64_small.npz"), **fx)
    print("wrote f64_small", {n: getattr(a, "dtype", None) for n, a in fx.items()})

    # VEBPR: `_fit_sgd_viewloss` is a fused-type function too (recom_vebpr.pyx:219) — float64 U / V train in double.
    # Purchases + views, 20 of the 80 users without views (the plain-BPR fallback branch), odd k.
    import importlib

    RefPV = importlib.import_module("cornac.data").PurchaseViewDataset
    RefVEBPR = importlib.import_module("cornac.models.bpr.recom_vebpr").VEBPR
    nu, ni, n_p, n_v, nu_view, k, epochs, lr, reg, alpha, mseed = 80, 60, 1200, 1500, 60, 9, 6, 0.05, 0.01, 0.4, 23
    pu, pi, _ = synth_pairs(nu, ni, n_p, 0.6, mseed)
    vu, vi, _ = synth_pairs(nu_view, ni, n_v, 0.4, mseed + 100)
    ds = RefPV.build([(int(a), int(b), 1.0) for a, b in zip(pu, pi)], [(int(a), int(b), 1.0) for a, b in zip(vu, vi)], seed=1)
    rs = np.random.RandomState(6)
    init = {"U": (rs.rand(ds.num_users, k) - 0.5) / k, "V": (rs.rand(ds.num_items, k) - 0.5) / k}
    m = RefVEBPR(k=k, max_iter=epochs, learning_rate=lr, lambda_reg=reg, alpha=alpha, seed=mseed,
                 init_params={n: a.copy() for n, a in init.items()}).fit(ds)
    assert m.u_factor.dtype == np.float64 and m.i_factor.dtype == np.float64
    np.savez_compressed(os.path.join(HERE, "vebpr_f64.npz"), pu=pu, pi=pi, vu=vu, vi=vi, k=np.int64(k), epochs=np.int64(epochs),
                        lr=np.float64(lr), reg=np.float64(reg), alpha=np.float64(alpha), seed=np.int64(mseed),
                        init_U=init["U"], init_V=init["V"], U=m.u_factor.copy(), V=m.i_factor.copy(),
                        score_0_3=np.float64(m.score(0, 3)))
    # (the reference's own score(user) allocates a float32 output for the float64 tables and fails in fast_dot with
    # "Buffer dtype mismatch, expected 'double' but got 'float'", recom_vebpr.pyx:356-357; score(user, item) works)
    try:
        m.score(0)
        raise SystemExit("the reference's VEBPR.score(user) was expected to fail on float64 tables")
    except ValueError as e:
        print("wrote vebpr_f64; reference score(user) on float64 tables:", e)


if __name__ == "__main__":
    main()
