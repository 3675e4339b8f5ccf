"""Validates the oracle against the LIVE reference (compiled from /root/reference into
oracle/_ref by oracle/build_ref.py).  Only runs where the reference exists (the build container);
skipped on the GPU box, which relies on the committed golden vectors instead."""
import numpy as np
import pytest


def assert_close(a, b, atol=2e-6, rtol=5e-6):
    """oracle (strict IEEE) vs reference (-ffast-math): a few ulp of the value magnitude"""
    err = np.abs(np.asarray(a) - np.asarray(b)).max()
    assert err <= atol + rtol * np.abs(b).max(), err

from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference not available / oracle/_ref not built")


def _pairs(nu, ni, nnz, seed):
    rs = np.random.RandomState(seed)
    keys = rs.permutation(np.unique(rs.randint(nu, size=3 * nnz).astype(np.int64) * ni + rs.randint(ni, size=3 * nnz)))[:nnz]
    return [(int(k // ni), int(k % ni), float(rs.randint(1, 6))) for k in keys]


@pytest.mark.parametrize("k,use_bias,seed", [(5, True, 1), (16, False, 2), (33, True, 3)])
def test_seeded_models_match_live_reference(oracle, k, use_bias, seed):
    ns = ref_loader.load()
    data = _pairs(80, 60, 1500, seed)
    ds = ns.Dataset.from_uir(data, seed=1)
    kw = dict(k=k, max_iter=6, learning_rate=0.03, lambda_reg=0.01, use_bias=use_bias, seed=seed)
    for ref_cls, or_cls in ((ns.BPR, oracle.BPROracle), (ns.WBPR, oracle.WBPROracle)):
        m, o = ref_cls(**kw).fit(ds), or_cls(**kw).fit(ds)
        assert_close(m.u_factors, o.u_factors)
        assert_close(m.i_factors, o.i_factors)
        assert_close(m.i_biases, o.i_biases)
    m, o = ns.MF(**kw).fit(ds), oracle.MFOracle(**kw).fit(ds)
    assert_close(m.u_factors, o.u_factors)
    assert_close(m.i_factors, o.i_factors)
    assert_close(m.i_biases, o.i_biases)
    assert np.abs(m.score(3) - o.score(3)).max() < 5e-6


def test_first_example_configuration_full_length(oracle):
    """BASELINE.json configs[0] (examples/first_example.py): BPR(k=10, max_iter=200, lr=0.001, reg=0.01, seed=123) and
    MF(k=10, max_iter=25, lr=0.01, reg=0.02, use_bias, seed=123) on ML-100K-shaped data (943 x 1682, 80 000 ratings) —
    16 M sequential BPR updates through the live reference and through the oracle: agreement stays at the ulp level
    (measured 4.5e-8 on factors of magnitude 0.1, 2.4e-7 on biases of magnitude 2.8)"""
    ns = ref_loader.load()
    rs = np.random.RandomState(1)
    p = 1.0 / np.arange(1, 1683) ** 0.8
    keys = np.unique(rs.randint(943, size=200000).astype(np.int64) * 1682 + rs.choice(1682, 200000, p=p / p.sum()))
    keys = rs.permutation(keys)[:80000]
    ds = ns.Dataset.from_uir([(int(k // 1682), int(k % 1682), float(rs.randint(1, 6))) for k in keys], seed=123)
    kw = dict(k=10, max_iter=200, learning_rate=0.001, lambda_reg=0.01, seed=123)
    m, o = ns.BPR(**kw).fit(ds), oracle.BPROracle(**kw).fit(ds)
    for name in ("u_factors", "i_factors", "i_biases"):
        assert_close(getattr(m, name), getattr(o, name), atol=1e-7, rtol=2e-7)
    kw = dict(k=10, max_iter=25, learning_rate=0.01, lambda_reg=0.02, use_bias=True, seed=123)
    m, o = ns.MF(**kw).fit(ds), oracle.MFOracle(**kw).fit(ds)
    for name in ("u_factors", "i_factors", "i_biases", "u_biases"):
        assert_close(getattr(m, name), getattr(o, name), atol=1e-7, rtol=1e-6)


def test_host_mirror_matches_reference_dataset_and_rank(oracle):
    """cornac_amd.Dataset builds the same arrays as cornac.data.Dataset, and the oracle's pinned
    rank() agrees with Recommender.rank on a tie-free score vector."""
    from cornac_amd import Dataset

    ns = ref_loader.load()
    data = _pairs(50, 40, 700, 9)
    a, b = ns.Dataset.from_uir(data, seed=1), Dataset.from_uir(data, seed=1)
    for x, y in zip(a.uir_tuple, b.uir_tuple):
        assert x.dtype == y.dtype and np.array_equal(x, y)
    assert np.array_equal(a.matrix.indptr, b.matrix.indptr) and np.array_equal(a.matrix.indices, b.matrix.indices)
    assert a.matrix.indices.dtype == b.matrix.indices.dtype == np.int32
    assert (a.num_users, a.num_items, a.global_mean) == (b.num_users, b.num_items, b.global_mean)
    m = ns.BPR(k=8, max_iter=5, seed=3).fit(a)
    s = m.score(7)
    cand = np.arange(5, 35)
    for k in (-1, 6):
        ref_rank, ref_scores = m.rank(7, item_indices=cand, k=k)
        ranked, scores = oracle.rank(s, a.num_items, len(a.iid_map), item_indices=cand, k=k)
        assert np.array_equal(scores, ref_scores)
        n = len(cand) if k == -1 else k
        assert np.array_equal(ranked[:n], ref_rank[:n])


def test_metric_mirrors_and_batched_eval_logic_match_reference():
    """cornac_amd.metrics == cornac.metrics on random rankings; cornac_amd.eval.ranking_eval (driven by
    a host stand-in model with rank()/rank_batch()) == the reference's ranking_eval."""
    import cornac_amd.eval as ev
    import cornac_amd.metrics as mm

    ns = ref_loader.load()
    rm = ns.metrics
    rs = np.random.RandomState(0)
    for _ in range(20):
        pd_rank = rs.permutation(50)
        gt_pos = rs.choice(50, rs.randint(1, 8), replace=False)
        for k in (1, 5, 10, -1):
            for mine, ref in ((mm.Precision, rm.Precision), (mm.Recall, rm.Recall), (mm.NDCG, rm.NDCG),
                              (mm.HitRatio, rm.HitRatio)):
                assert mine(k=k).compute(gt_pos=gt_pos, pd_rank=pd_rank) == pytest.approx(
                    ref(k=k).compute(gt_pos=gt_pos, pd_rank=pd_rank))

            for mine, ref in ((mm.FMeasure, rm.FMeasure), (mm.NCRR, rm.NCRR)):
                assert mine(k=k).compute(gt_pos=gt_pos, pd_rank=pd_rank[:30]) == pytest.approx(
                    ref(k=k).compute(gt_pos=gt_pos, pd_rank=pd_rank[:30]))
        cand = np.sort(rs.choice(50, 35, replace=False))
        gt_in = rs.choice(cand, 3, replace=False)
        sc = (np.round(rs.normal(size=35) * 2) / 2).astype(np.float32)          # many tied scores
        kw = dict(item_indices=cand, pd_scores=sc, gt_pos=gt_in, gt_neg=np.setdiff1d(cand, gt_in),
                  pd_rank=cand[np.argsort(sc, kind="stable")[::-1]])
        for mine, ref in ((mm.AUC, rm.AUC), (mm.MAP, rm.MAP), (mm.MRR, rm.MRR)):
            assert mine().compute(**kw) == pytest.approx(ref().compute(**kw), rel=1e-12)

    class HostModel:  # scores from a fixed matrix; same tie rule as the device kernels
        def __init__(self, S, n_items):
            self.S, self.num_items, self.total_items = S, n_items, n_items

        def rank(self, user_idx, item_indices=None, k=-1, **kw):
            s = self.S[user_idx]
            item_indices = np.arange(self.num_items) if item_indices is None else np.asarray(item_indices)
            sc = s[item_indices]
            order = np.argsort(sc, kind="stable")[::-1]
            return item_indices[order], sc      # like Recommender.rank: every candidate, the first k (here all) in order

        def rank_batch(self, users, k=10, exclude=None):
            width = self.num_items if k == -1 else k
            out = np.full((len(users), width), -1, np.int32)
            out_s = np.full((len(users), width), -np.inf, np.float32)
            for r, u in enumerate(users):
                ex = exclude[1][exclude[0][r]:exclude[0][r + 1]]
                cand = np.setdiff1d(np.arange(self.num_items), ex)
                rr = self.rank(u, cand, k)[0][:width]
                out[r, :len(rr)] = rr
                out_s[r, :len(rr)] = self.S[u][rr]
            return out, out_s

    class PositionsModel(HostModel):  # + the counting interface of the device scorer (cornac_hip_rank_positions)
        calls = 0

        def rank_positions_batch(self, users, targets, exclude=None):
            PositionsModel.calls += 1
            out = [[], [], [], []]
            for r, u in enumerate(users):
                ex = exclude[1][exclude[0][r]:exclude[0][r + 1]]
                cand = np.setdiff1d(np.arange(self.num_items), ex)
                sc = self.S[u][cand]
                for t in targets[1][targets[0][r]:targets[0][r + 1]]:
                    s = self.S[u][t]
                    out[0].append(int((sc > s).sum()))
                    out[1].append(int((sc > s).sum() + ((sc == s) & (cand > t)).sum()))
                    out[2].append(int((sc >= s).sum()))
                    out[3].append(s)
            return (np.array(out[0], np.int32), np.array(out[1], np.int32), np.array(out[2], np.int32),
                    np.array(out[3], np.float32))

    data = _pairs(60, 45, 1200, 4)
    rs.shuffle(data)
    train = ns.Dataset.build(data[:900])
    test = ns.Dataset.build(data[900:], global_uid_map=train.uid_map, global_iid_map=train.iid_map,
                            exclude_unknowns=True)
    S = rs.normal(size=(train.num_users, train.num_items)).astype(np.float32)
    model = HostModel(S, train.num_items)
    ref_metrics = [rm.Recall(k=10), rm.NDCG(k=10), rm.Precision(k=5)]
    my_metrics = [mm.Recall(k=10), mm.NDCG(k=10), mm.Precision(k=5)]
    ref_avg, ref_user = ns.eval_methods.base_method.ranking_eval(model, ref_metrics, train, test)
    avg, user = ev.ranking_eval(model, my_metrics, train, test)
    assert np.allclose(avg, ref_avg) and user[0].keys() == ref_user[0].keys()
    avg2, _ = ev.ranking_eval(model, ref_metrics + [rm.AUC()], train, test)  # k = -1 -> per-user full-rank flow
    ref_avg2, _ = ns.eval_methods.base_method.ranking_eval(model, ref_metrics + [rm.AUC()], train, test)
    assert np.allclose(avg2, ref_avg2)
    # full-list metrics through the batched flow (mirrors with compute_full_batch), tied scores included
    model_t = HostModel((np.round(S * 2) / 2).astype(np.float32), train.num_items)
    names = ("NDCG", "Recall", "Precision", "NCRR", "FMeasure", "HitRatio")
    sets = (([rm.AUC(), rm.MAP(), rm.MRR()], [mm.AUC(), mm.MAP(), mm.MRR()]),
            ([getattr(rm, n)(k=-1) for n in names] + [rm.AUC()], [getattr(mm, n)(k=-1) for n in names] + [mm.AUC()]),
            ([getattr(rm, n)(k=-1) for n in names] + [rm.Recall(k=10)],          # mixed: rank() is asked for 10 items
             [getattr(mm, n)(k=-1) for n in names] + [mm.Recall(k=10)]),
            ([rm.AUC(), rm.MAP(), rm.NCRR(k=10), rm.FMeasure(k=5), rm.Recall(k=10), rm.NDCG(k=3)],
             [mm.AUC(), mm.MAP(), mm.NCRR(k=10), mm.FMeasure(k=5), mm.Recall(k=10), mm.NDCG(k=3)]))
    for mdl in (model, model_t, PositionsModel(model.S, train.num_items), PositionsModel(model_t.S, train.num_items)):
        for ref_all, my_all in sets:
            ref_avg3, ref_user3 = ns.eval_methods.base_method.ranking_eval(mdl, ref_all, train, test)
            avg3, user3 = ev.ranking_eval(mdl, my_all, train, test, batch_users_full=16, batch_users=16)
            assert np.allclose(avg3, ref_avg3, rtol=1e-9), (avg3, ref_avg3)
            for mine_u, ref_u in zip(user3, ref_user3):
                assert mine_u.keys() == ref_u.keys()
                assert np.allclose([mine_u[u] for u in ref_u], [ref_u[u] for u in ref_u], rtol=1e-9)
    assert PositionsModel.calls >= 24   # every metric list above went through the counting interface
    # mixed with @k metrics the reference asks rank() for max_k items but receives every candidate: MRR still sees the list
    for mdl in (model, PositionsModel(S, train.num_items)):
        ref_avg4, _ = ns.eval_methods.base_method.ranking_eval(mdl, [rm.MRR(), rm.Recall(k=3)], train, test)
        avg4, _ = ev.ranking_eval(mdl, [mm.MRR(), mm.Recall(k=3)], train, test)
        assert np.allclose(avg4, ref_avg4, rtol=1e-9)


@pytest.mark.parametrize("opt", ["sgd", "adam", "rmsprop", "adagrad"])
def test_mf_minibatch_oracle_matches_live_reference(opt):
    """MF(backend="pytorch") of the real reference vs oracle/mf_minibatch_oracle.py on a fresh random case"""
    from cornac_amd import Dataset
    from oracle import mf_minibatch_oracle

    ns = ref_loader.load()
    data = _pairs(70, 45, 700, 9)
    ref_ds = ns.Dataset.from_uir(data, seed=11)
    m = ns.MF(k=7, backend="pytorch", optimizer=opt, max_iter=2, batch_size=48, learning_rate=0.03, lambda_reg=0.01,
              use_bias=True, seed=4, verbose=False).fit(ref_ds)
    ds = Dataset.from_uir(data, seed=11)
    ds.reset()
    rng = np.random.RandomState(4)
    U = rng.normal(0.0, 0.01, (ds.num_users, 7)).astype(np.float32)
    V = rng.normal(0.0, 0.01, (ds.num_items, 7)).astype(np.float32)
    rid, cid, val = ds.uir_tuple
    batches = []
    for _ in range(2):
        batches += list(ds.idx_iter(len(val), 48, shuffle=True))
    got = mf_minibatch_oracle.fit(U, V, np.zeros(ds.num_users), np.zeros(ds.num_items), np.float32(ds.global_mean), rid, cid,
                                  val.astype(np.float32), batches, opt, 0.03, 0.01, True)
    for a, b in zip(got[:4], (m.u_factors, m.i_factors, m.u_biases, m.i_biases)):
        assert_close(a, np.asarray(b))


@pytest.mark.parametrize("opt,use_bias,p", [("sgd", True, 0.3), ("adam", True, 0.5), ("adagrad", False, 0.2), ("rmsprop", True, 0.9),
                                            ("sgd", True, 1.0)])
def test_mf_minibatch_dropout_matches_live_reference(opt, use_bias, p):
    """MF(backend="pytorch", dropout=p) of the real reference on a fresh random case vs the oracle fed with the masks
    `mf_minibatch_oracle.dropout_masks` restates (torch's CPU generator after manual_seed and the embedding
    initialisations; user rows, then item rows, per batch).  p = 1 drops everything without drawing."""
    from cornac_amd import Dataset
    from oracle import mf_minibatch_oracle

    ns = ref_loader.load()
    data = _pairs(70, 45, 700, 9)
    ref_ds = ns.Dataset.from_uir(data, seed=11)
    m = ns.MF(k=7, backend="pytorch", optimizer=opt, max_iter=2, batch_size=48, learning_rate=0.03, lambda_reg=0.01,
              use_bias=use_bias, dropout=p, seed=4, verbose=False).fit(ref_ds)
    ds = Dataset.from_uir(data, seed=11)
    ds.reset()
    rng = np.random.RandomState(4)
    U = rng.normal(0.0, 0.01, (ds.num_users, 7)).astype(np.float32)
    V = rng.normal(0.0, 0.01, (ds.num_items, 7)).astype(np.float32)
    rid, cid, val = ds.uir_tuple
    batches = []
    for _ in range(2):
        batches += list(ds.idx_iter(len(val), 48, shuffle=True))
    if p < 1.0:
        keep, scale = mf_minibatch_oracle.dropout_masks(p, 4, ds.num_users, ds.num_items, 7, use_bias, [len(b) for b in batches])
    else:
        keep, scale = [(np.zeros((len(b), 7), np.uint8),) * 2 for b in batches], 0.0
    got = mf_minibatch_oracle.fit(U, V, np.zeros(ds.num_users), np.zeros(ds.num_items), np.float32(ds.global_mean), rid, cid,
                                  val.astype(np.float32), batches, opt, 0.03, 0.01, use_bias, keep=keep, keep_scale=scale)
    for a, b in zip(got[:4] if use_bias else got[:2], (m.u_factors, m.i_factors, m.u_biases, m.i_biases)):
        assert_close(a, np.asarray(b))


def _f64_init(nu, ni, k, seed):
    rs = np.random.RandomState(seed)
    return {"U": (rs.rand(nu, k) - 0.5) / k, "V": (rs.rand(ni, k) - 0.5) / k, "Bi": 0.01 * rs.randn(ni)}


@pytest.mark.parametrize("k,seed", [(5, 1), (16, 2), (33, 3)])
def test_float64_models_match_live_reference(oracle, k, seed):
    """`_fit_sgd` is a fused-type function (recom_bpr.pyx:211-214): float64 init_params train in double.  The oracle's
    float64 restatement against the live reference, BPR and WBPR, score() included; a dtype mix raises on both sides."""
    ns = ref_loader.load()
    ds = ns.Dataset.from_uir(_pairs(80, 60, 1500, seed), seed=1)
    kw = dict(k=k, max_iter=6, learning_rate=0.03, lambda_reg=0.01, seed=seed)
    for ref_cls, or_cls in ((ns.BPR, oracle.BPROracle), (ns.WBPR, oracle.WBPROracle)):
        ip = _f64_init(ds.num_users, ds.num_items, k, seed)
        m = ref_cls(init_params={n: a.copy() for n, a in ip.items()}, **kw).fit(ds)
        o = or_cls(init_params={n: a.copy() for n, a in ip.items()}, **kw).fit(ds)
        assert m.u_factors.dtype == np.float64 and o.u_factors.dtype == np.float64
        for name in ("u_factors", "i_factors", "i_biases"):
            assert np.abs(getattr(m, name) - getattr(o, name)).max() < 1e-13, name
        assert np.abs(m.score(3) - o.score(3)).max() < 1e-13
        assert np.abs(m.u_factors - ip["U"]).max() > 1e-4, "the run must have trained"
    ip = _f64_init(ds.num_users, ds.num_items, k, seed)
    ip["V"] = ip["V"].astype(np.float32)
    with pytest.raises(ValueError):
        ns.BPR(init_params=dict(ip), **kw).fit(ds)
    with pytest.raises(ValueError):
        oracle.BPROracle(init_params=dict(ip), **kw).fit(ds)


@pytest.mark.parametrize("given_init", [False, True])
def test_wmf_oracle_and_host_class_match_the_reference_wmf_code(monkeypatch, given_init):
    """The reference's OWN WMF (cornac/models/wmf/recom_wmf.py + wmf.py, unmodified: xavier init from the seed, item_iter
    shuffling, batch_C construction, the TF1 graph, sess.run per batch) executed over oracle/tf1_shim — torch forward and
    autograd of the loss the reference's code builds; only the gather gradient / clip / TF1 Adam rules are restated
    there — against cornac_amd.WMF whose device layer is the WMF oracle (tests/fake_device.py).  Same U, V to float32
    rounding over 3 epochs of 4 batches, same scores; so the oracle's hand-derived gradients, its Adam and the host class's
    loop are the reference's.  (With a real TensorFlow importable the same test runs against it.)"""
    import fake_device
    from oracle import ref_wmf

    import cornac_amd as ca

    RefWMF = ref_wmf.load_wmf()
    ns = ref_loader.load()
    fake_device.install(monkeypatch)
    rs = np.random.RandomState(4)
    keys = rs.permutation(70 * 50)[:900]
    data = [("u%d" % (k // 50), "i%d" % (k % 50), float(rs.randint(1, 6))) for k in keys]
    build = lambda: ns.Dataset.from_uir(data, seed=11)   # noqa: E731 — item_iter shuffles with the data set's own generator
    ds = build()
    kw = dict(k=6, max_iter=3, batch_size=16, learning_rate=0.01, lambda_u=0.02, lambda_v=0.03, a=1.0, b=0.05, seed=7,
              verbose=False)
    init = None
    if given_init:
        init = {"U": rs.normal(0, 0.1, (ds.num_users, 6)).astype(np.float32),
                "V": rs.normal(0, 0.1, (ds.num_items, 6)).astype(np.float32)}
    theirs = RefWMF(init_params=None if init is None else {n: a.copy() for n, a in init.items()}, **kw).fit(build())
    ours = ca.WMF(init_params=None if init is None else {n: a.copy() for n, a in init.items()}, **kw).fit(build())
    assert np.abs(theirs.U).max() > 0.05 and (init is None or np.abs(theirs.U - init["U"]).max() > 0.01)
    assert np.abs(ours.U - theirs.U).max() < 5e-6, np.abs(ours.U - theirs.U).max()
    assert np.abs(ours.V - theirs.V).max() < 5e-6, np.abs(ours.V - theirs.V).max()
    assert np.abs(ours.score(3) - theirs.score(3)).max() < 1e-5
    assert abs(ours.score(3, 5) - theirs.score(3, 5)) < 1e-5
