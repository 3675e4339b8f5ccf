#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REAL REFERENCE (only possible in the build
container, where /root/reference exists and oracle/build_ref.py has compiled its hot-path
extensions into oracle/_ref/).  The fixtures travel to the GPU box; this script does not.

Each fixture stores the exact inputs (interaction triplets in insertion order) and what the
compiled reference (Cython BPR/WBPR/MF, seed => num_threads = 1) learned from them, plus
`score()` / `rank()` outputs of the fitted reference model for a few users.

    python tests/golden/make_golden.py
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from oracle import build_ref, ref_loader  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def synth_pairs(n_users, n_items, nnz, a, seed):
    rs = np.random.RandomState(seed)
    p = 1.0 / np.arange(1, n_items + 1) ** a
    p /= p.sum()
    pairs = set()
    while len(pairs) < nnz:
        pairs.add((int(rs.randint(n_users)), int(rs.choice(n_items, p=p))))
    pairs = sorted(pairs)
    ratings = rs.randint(1, 6, size=len(pairs)).astype(np.float64)
    # shuffle the insertion order so COO order != CSR order (exercises MF's order dependence)
    order = rs.permutation(len(pairs))
    u = np.array([pairs[t][0] for t in order], np.int64)
    i = np.array([pairs[t][1] for t in order], np.int64)
    return u, i, ratings[order]


def dataset(ns, u, i, r):
    return ns.Dataset.from_uir([(int(a), int(b), float(c)) for a, b, c in zip(u, i, r)], seed=123)


def rank_samples(model, users, k):
    out = {}
    for t, uu in enumerate(users):
        ranked, scores = model.rank(int(uu), k=-1)
        out["rank_full_%d" % t] = ranked.astype(np.int64)
        out["rank_scores_%d" % t] = scores.astype(np.float32)
        out["score_%d" % t] = model.score(int(uu)).astype(np.float32)
        topk, _ = model.rank(int(uu), k=k)
        out["rank_top_%d" % t] = topk[:k].astype(np.int64)
    out["rank_users"] = np.asarray(users, np.int64)
    out["rank_k"] = np.int64(k)
    return out


def main():
    build_ref.build()
    ns = ref_loader.load()
    cases = {
        # name: (n_users, n_items, nnz, zipf, data_seed, k, epochs, lr, reg, model_seed)
        "tiny": (12, 9, 40, 0.5, 3, 4, 30, 0.05, 0.01, 123),
        "small": (60, 40, 600, 0.8, 5, 8, 10, 0.05, 0.01, 42),
        "odd_k": (70, 50, 900, 0.8, 7, 10, 5, 0.01, 0.02, 7),
        "ml100k_shape": (943, 1682, 20000, 0.8, 1, 10, 3, 0.001, 0.01, 123),
    }
    for name, (nu, ni, nnz, a, dseed, k, epochs, lr, reg, mseed) in cases.items():
        u, i, r = synth_pairs(nu, ni, nnz, a, dseed)
        ds = dataset(ns, u, i, r)
        fx = {"users": u, "items": i, "ratings": r, "k": np.int64(k), "epochs": np.int64(epochs),
              "lr": np.float64(lr), "reg": np.float64(reg), "seed": np.int64(mseed)}
        rank_users = [0, nu // 2, nu - 1]
        for tag, cls in (("bpr", ns.BPR), ("wbpr", ns.WBPR)):
            for use_bias in (True, False):
                if tag == "wbpr" and not use_bias:
                    continue
                m = cls(k=k, max_iter=epochs, learning_rate=lr, lambda_reg=reg, use_bias=use_bias, seed=mseed).fit(ds)
                sfx = "" if use_bias else "_nobias"
                fx[tag + sfx + "_U"] = m.u_factors.copy()
                fx[tag + sfx + "_V"] = m.i_factors.copy()
                fx[tag + sfx + "_B"] = m.i_biases.copy()
                if tag == "bpr" and use_bias:
                    for kk, vv in rank_samples(m, rank_users, 5).items():
                        fx["bpr_" + kk] = vv
        m = ns.MF(k=k, max_iter=epochs, learning_rate=lr, lambda_reg=reg * 2, use_bias=True, seed=mseed).fit(ds)
        fx["mf_U"], fx["mf_V"] = m.u_factors.copy(), m.i_factors.copy()
        fx["mf_Bu"], fx["mf_Bi"] = m.u_biases.copy(), m.i_biases.copy()
        fx["mf_mu"] = np.float32(m.global_mean)
        for kk, vv in rank_samples(m, rank_users, 5).items():
            fx["mf_" + kk] = vv
        m = ns.MF(k=k, max_iter=epochs, learning_rate=lr, lambda_reg=reg * 2, use_bias=False, seed=mseed).fit(ds)
        fx["mf_nobias_U"], fx["mf_nobias_V"] = m.u_factors.copy(), m.i_factors.copy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
        print("wrote", name, {kk: getattr(vv, "shape", None) for kk, vv in list(fx.items())[:6]})

    # VEBPR: purchases + views (some users without views -> BPR fallback branch), real reference
    import importlib

    RefPV = importlib.import_module("cornac.data").PurchaseViewDataset
    RefVEBPR = importlib.import_module("cornac.models.bpr.recom_vebpr").VEBPR
    for name, (nu, ni, n_p, n_v, nu_view, k, epochs, lr, reg, alpha, mseed) in {
        "vebpr_small": (80, 60, 1200, 1500, 60, 8, 8, 0.05, 0.01, 0.5, 7),
        "vebpr_odd": (150, 90, 2500, 2500, 150, 13, 5, 0.02, 0.05, 0.3, 11),
    }.items():
        pu, pi, _ = synth_pairs(nu, ni, n_p, 0.6, mseed)
        vu, vi, _ = synth_pairs(nu_view, ni, n_v, 0.4, mseed + 100)
        pur = [(int(a), int(b), 1.0) for a, b in zip(pu, pi)]
        view = [(int(a), int(b), 1.0) for a, b in zip(vu, vi)]
        ds = RefPV.build(pur, view, seed=1)
        m = RefVEBPR(k=k, max_iter=epochs, learning_rate=lr, lambda_reg=reg, alpha=alpha, seed=mseed).fit(ds)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), pu=pu, pi=pi, vu=vu, vi=vi, k=np.int64(k),
                            epochs=np.int64(epochs), lr=np.float64(lr), reg=np.float64(reg), alpha=np.float64(alpha),
                            seed=np.int64(mseed), U=m.u_factor.copy(), V=m.i_factor.copy(),
                            score0=m.score(0).astype(np.float32))
        print("wrote", name)

    # VBPR (torch autograd + Adam in the reference): small case with visual features
    import types

    RefVBPR = importlib.import_module("cornac.models.vbpr").VBPR
    rs = np.random.RandomState(3)
    vu, vi, vr = synth_pairs(40, 30, 400, 0.5, 21)
    vr = (1 + (vu + vi) % 3).astype(np.float64)
    feats = rs.uniform(0, 1, (30, 50)).astype(np.float32)
    ds = ns.Dataset.from_uir([(int(a), int(b), float(c)) for a, b, c in zip(vu, vi, vr)], seed=5)
    ds.item_image = types.SimpleNamespace(features=feats)
    kw = dict(k=6, k2=5, n_epochs=4, batch_size=50, learning_rate=0.01, lambda_w=0.01, lambda_b=0.01, lambda_e=0.001,
              seed=9)
    m = RefVBPR(verbose=False, **kw).fit(ds)
    np.savez_compressed(os.path.join(OUT, "vbpr_small.npz"), users=vu, items=vi, ratings=vr, features=feats,
                        Bi=m.beta_item, Gu=m.gamma_user, Gi=m.gamma_item, Tu=m.theta_user, E=m.emb_matrix,
                        Bp=m.beta_prime, theta_item=m.theta_item, visual_bias=m.visual_bias,
                        score0=m.score(0).astype(np.float32), **{k_: np.float64(v_) for k_, v_ in kw.items()})
    print("wrote vbpr_small")

    # the reference's own known-answer test for this path: tests/cornac/utils/test_fastdot.py:26-37
    vec = np.array([1, 2], np.float32)
    mat = np.array([[1, 2], [3, 4]], np.float32)
    out = np.array([0, 0], np.float32)
    ns.fast_dot(vec, mat, out)
    rs = np.random.RandomState(0)
    v2 = rs.normal(size=64).astype(np.float32)
    m2 = rs.normal(size=(257, 64)).astype(np.float32)
    o2 = rs.normal(size=257).astype(np.float32)
    o2_in = o2.copy()
    ns.fast_dot(v2, m2, o2)
    np.savez_compressed(os.path.join(OUT, "fast_dot.npz"), kat_vec=vec, kat_mat=mat, kat_out=out, vec=v2, mat=m2,
                        out_in=o2_in, out=o2)
    print("wrote fast_dot", out)



def make_wmf_fixture():
    """WMF: the reference cannot run here (TensorFlow absent) -> fixture from oracle/wmf_oracle.py, with the
    reference's host-side iterator protocol (Dataset.item_iter after train_set.reset())."""
    sys.path.insert(0, ROOT)
    from cornac_amd.data import Dataset
    from oracle.wmf_oracle import WmfOracle

    rs = np.random.RandomState(11)
    nu, ni, k, nnz, bs, iters = 150, 70, 12, 1500, 32, 4
    keys = rs.permutation(nu * ni)[:nnz]
    u, i = keys // ni, keys % ni
    r = rs.randint(1, 6, nnz).astype(np.float32)
    ds = Dataset.from_uir([(int(a), int(b), float(c)) for a, b, c in zip(u, i, r)], seed=123)
    ds.reset()
    U0 = rs.uniform(-0.2, 0.2, (ds.num_users, k)).astype(np.float32)
    V0 = rs.uniform(-0.2, 0.2, (ds.num_items, k)).astype(np.float32)
    hp = dict(lambda_u=0.02, lambda_v=0.03, a=1.0, b=0.01, lr=0.005)
    o = WmfOracle(U0, V0, ds.csc_matrix, **hp)
    batches = []
    for _ in range(iters):
        batches += list(ds.item_iter(bs, shuffle=True))
    losses = o.fit_batches(batches)
    ptr = np.zeros(len(batches) + 1, np.int64)
    np.cumsum([len(b) for b in batches], out=ptr[1:])
    u, i = ds.uir_tuple[0], ds.uir_tuple[1]  # mapped ids: identity under a rebuild with from_uir
    np.savez_compressed(os.path.join(OUT, "wmf_small.npz"), users=u, items=i, ratings=r, n_users=ds.num_users,
                        n_items=ds.num_items, k=k, batch_size=bs, max_iter=iters, U0=U0, V0=V0, U=o.U, V=o.V,
                        losses=np.array(losses), batch_ids=np.concatenate(batches), batch_ptr=ptr, **hp)
    print("wrote wmf_small")


def make_mf_minibatch_fixture():
    """MF(backend="pytorch", optimizer=...) of the REAL reference on CPU torch, all four optimisers."""
    build_ref.build()
    ns = ref_loader.load()
    u, i, r = synth_pairs(90, 50, 1100, 0.7, 21)
    ds = dataset(ns, u, i, r)
    fx = {"users": u, "items": i, "ratings": r, "k": np.int64(6), "epochs": np.int64(3), "batch_size": np.int64(64),
          "lr": np.float64(0.02), "reg": np.float64(0.03), "seed": np.int64(5)}
    for opt in ("sgd", "adam", "rmsprop", "adagrad"):
        for use_bias in (True, False):
            m = ns.MF(k=6, backend="pytorch", optimizer=opt, max_iter=3, batch_size=64, learning_rate=0.02,
                      lambda_reg=0.03, use_bias=use_bias, seed=5).fit(ds)
            tag = opt + ("" if use_bias else "_nobias")
            fx[tag + "_U"], fx[tag + "_V"] = m.u_factors.copy(), m.i_factors.copy()
            fx[tag + "_Bu"], fx[tag + "_Bi"] = np.asarray(m.u_biases).copy(), np.asarray(m.i_biases).copy()
    np.savez_compressed(os.path.join(OUT, "mf_minibatch.npz"), **fx)
    print("wrote mf_minibatch")


def make_mf_dropout_fixture():
    """MF(backend="pytorch", dropout=p) of the REAL reference on CPU torch: the dropout masks come from torch's CPU
    generator (seeded by recom_mf.py:221-222), so the fixture also pins how cornac_amd/mf.py restates their draw."""
    build_ref.build()
    ns = ref_loader.load()
    u, i, r = synth_pairs(90, 50, 1100, 0.7, 21)
    ds = dataset(ns, u, i, r)
    fx = {"users": u, "items": i, "ratings": r, "k": np.int64(6), "epochs": np.int64(3), "batch_size": np.int64(64),
          "lr": np.float64(0.02), "reg": np.float64(0.03), "seed": np.int64(5)}
    for opt, use_bias, p in (("sgd", True, 0.3), ("adam", True, 0.5), ("rmsprop", False, 0.2), ("adagrad", True, 0.1)):
        m = ns.MF(k=6, backend="pytorch", optimizer=opt, max_iter=3, batch_size=64, learning_rate=0.02, lambda_reg=0.03,
                  use_bias=use_bias, dropout=p, seed=5, verbose=False).fit(ds)
        tag = "%s_p%d%s" % (opt, round(100 * p), "" if use_bias else "_nobias")
        fx[tag + "_U"], fx[tag + "_V"] = m.u_factors.copy(), m.i_factors.copy()
        fx[tag + "_Bu"], fx[tag + "_Bi"] = np.asarray(m.u_biases).copy(), np.asarray(m.i_biases).copy()
    np.savez_compressed(os.path.join(OUT, "mf_minibatch_dropout.npz"), **fx)
    print("wrote mf_minibatch_dropout")


if __name__ == "__main__":
    if "--mf-dropout-only" in sys.argv:
        make_mf_dropout_fixture()
    elif "--wmf-only" in sys.argv:
        make_wmf_fixture()
    elif "--mf-minibatch-only" in sys.argv:
        make_mf_minibatch_fixture()
    else:
        main()
        make_mf_minibatch_fixture()
        make_mf_dropout_fixture()
        make_wmf_fixture()
