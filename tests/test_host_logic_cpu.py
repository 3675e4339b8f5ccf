"""Host logic of the model classes on the CPU, with the device layer replaced by the oracle-backed doubles of
tests/fake_device.py: what `BPR / WBPR / MF .fit()` hand to the device (initial factors, the derived mt19937 seeds and
their order, mode selection) reproduces the REAL reference's learned parameters (golden vectors made by the compiled
reference, tests/golden/), and the Recommender / evaluation / experiment layers behave like the reference's around
those parameters.  The HIP kernels themselves are covered by the `-m gpu` tests through the real C ABI."""
import copy
import pickle

import numpy as np
import pytest

import fake_device
from conftest import golden_dataset, load_golden
from test_oracle_golden import CASES, _kw, assert_close


@pytest.fixture()
def device_double(monkeypatch, tmp_path):
    fake_device.install(monkeypatch)
    monkeypatch.chdir(tmp_path)     # the reference's Experiment drops its CornacExp-*.log report into the working directory


@pytest.mark.parametrize("name", CASES)
def test_seeded_fit_through_the_host_classes_reproduces_the_reference_goldens(device_double, name):
    from cornac_amd import BPR, MF, WBPR

    fx = load_golden(name)
    ds = golden_dataset(fx)
    for use_bias, sfx in ((True, ""), (False, "_nobias")):
        m = BPR(use_bias=use_bias, **_kw(fx)).fit(ds)
        assert m.effective_mode == "deterministic"
        for attr, key in (("u_factors", "_U"), ("i_factors", "_V"), ("i_biases", "_B")):
            assert_close(getattr(m, attr), fx["bpr" + sfx + key])
    w = WBPR(**_kw(fx)).fit(ds)
    for attr, key in (("u_factors", "wbpr_U"), ("i_factors", "wbpr_V"), ("i_biases", "wbpr_B")):
        assert_close(getattr(w, attr), fx[key])
    kw = _kw(fx)
    kw["lambda_reg"] *= 2
    mf = MF(use_bias=True, **kw).fit(ds)
    for attr, key in (("u_factors", "mf_U"), ("i_factors", "mf_V"), ("u_biases", "mf_Bu"), ("i_biases", "mf_Bi")):
        assert_close(getattr(mf, attr), fx[key])
    assert abs(float(mf.global_mean) - float(fx["mf_mu"])) < 1e-7
    mf = MF(use_bias=False, **kw).fit(ds)
    assert_close(mf.u_factors, fx["mf_nobias_U"])
    assert_close(mf.i_factors, fx["mf_nobias_V"])


def test_mode_selection_seed_protocol_and_warm_start(device_double):
    from cornac_amd import BPR, _lib

    fx = load_golden("small")
    ds = golden_dataset(fx)
    seen = []
    real = fake_device.FakeBprTrainer.fit_epochs

    def spy(self, n, lr, reg, use_bias, neg, mode, flags=0):
        seen.append((mode, list(self.calls)))
        return real(self, n, lr, reg, use_bias, neg, mode, flags)

    fake_device.FakeBprTrainer.fit_epochs = spy
    try:
        m = BPR(k=4, max_iter=2, seed=5).fit(ds)
        assert seen[-1][0] == _lib.MODE_DETERMINISTIC and seen[-1][1][0][0] == "mt19937"
        # recom_bpr.pyx:190-191 + :55-59: two draws from the model's generator AFTER the factor init, each turned into
        # thread 0's engine seed by RandomState(draw).randint(2**31)
        rng = np.random.RandomState(5)
        rng.uniform(0, 1, (len(ds.uid_map), 4))
        rng.uniform(0, 1, (len(ds.iid_map), 4))
        want = [int(np.random.RandomState(rng.randint(2 ** 31)).randint(2 ** 31)) for _ in range(2)]
        assert list(seen[-1][1][0][1:3]) == want and seen[-1][1][0][3] is False
        h = BPR(k=4, max_iter=2).fit(ds)               # no seed -> throughput mode
        assert seen[-1][0] == _lib.MODE_HOGWILD and seen[-1][1][0][0] == "hogwild" and np.isfinite(h.u_factors).all()
        assert BPR(k=4, max_iter=1, seed=5, mode="hogwild").fit(ds).effective_mode == "hogwild"
        # a second fit() continues from the trained factors and the advanced generator (the reference's warm start)
        first = m.u_factors.copy()
        with pytest.warns(UserWarning):
            m.fit(ds)
        assert not np.array_equal(first, m.u_factors) and len(seen[-1][1]) == 1
        fresh = m.clone().fit(ds)
        assert np.array_equal(fresh.u_factors, first)
        fixed = BPR(k=4, max_iter=3, seed=5, trainable=False, init_params={"U": first.copy()}).fit(ds)
        assert np.array_equal(fixed.u_factors, first)   # trainable=False: nothing is sent to the device
        with pytest.raises(ValueError):
            BPR(k=4, seed=1, init_params={"U": first.astype(np.float64)}).fit(ds)
    finally:
        fake_device.FakeBprTrainer.fit_epochs = real


def test_recommender_surface_around_fitted_parameters(device_double, tmp_path):
    from cornac_amd import MF, ScoreException

    fx = load_golden("small")
    ds = golden_dataset(fx)
    m = MF(k=6, max_iter=5, seed=3).fit(ds)
    s = m.score(2)
    assert s.dtype == np.float32 and len(s) == ds.num_items and m.score(2, 4) == pytest.approx(float(s[4]), abs=1e-6)
    with pytest.raises(ScoreException):
        m.score(2, ds.num_items + 3)
    cand = np.arange(3, 30)
    ranked, scores = m.rank(2, item_indices=cand, k=5)
    assert np.array_equal(scores, s[cand]) and list(ranked[:5]) == list(cand[np.argsort(s[cand], kind="stable")[::-1]][:5])
    assert len(ranked) == len(cand) and sorted(ranked.tolist()) == sorted(cand.tolist())  # recommender.py:521-528
    full, _ = m.rank(2)
    assert sorted(full.tolist()) == list(range(ds.num_items)) and np.all(np.diff(s[full]) <= 0)
    assert ds.min_rating <= m.rate(2, 4) <= ds.max_rating
    # recom_mf.py:281-286: an unknown user still gets the item's bias; an unknown item is a ScoreException -> global mean
    assert m.rate(ds.num_users + 1, 0) == pytest.approx(ds.global_mean + m.i_biases[0], abs=1e-6)
    assert m.rate(2, ds.num_items + 5) == pytest.approx(ds.global_mean)
    pairs_u, pairs_i = np.array([0, 2, 2, ds.num_users + 4]), np.array([1, 4, ds.num_items + 2, 0])
    got = m.rate_batch(pairs_u, pairs_i)
    want = [float(m.rate(int(u), int(i))) for u, i in zip(pairs_u, pairs_i)]
    assert np.allclose(got, want, atol=1e-6)
    uid = ds.user_ids[2]
    seen_items = {ds.item_ids[i] for i in ds.matrix[2].indices}
    assert not seen_items & set(m.recommend(uid, k=8, remove_seen=True, train_set=ds))
    assert m.recommend(uid, k=3) == [ds.item_ids[i] for i in full[:3]]
    with pytest.raises(ValueError):
        m.recommend("nobody")
    # persistence: the parameters travel, the device objects do not
    again = pickle.loads(pickle.dumps(copy.deepcopy(m)))
    assert np.array_equal(again.u_factors, m.u_factors) and np.array_equal(again.score(2), s)
    loaded = MF.load(m.save(str(tmp_path)))
    assert np.array_equal(loaded.rank(2, k=7)[0], m.rank(2, k=7)[0])
    c = m.clone({"k": 9})
    assert c.k == 9 and c.max_iter == 5 and c.u_factors is None


def test_first_example_flow_equals_the_reference_flow(device_double):
    """examples/first_example.py end to end on the CPU: the reference's RatioSplit + Experiment over MF and BPR against
    cornac_amd's with the device double — same splits, same learned parameters (to the oracle's ulp-level agreement
    with the compiled reference), hence the same report"""
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ns = ref_loader.load()
    import importlib

    from cornac_amd import BPR, MF, Experiment, RatioSplit
    from cornac_amd import metrics as mm

    RefExperiment = importlib.import_module("cornac.experiment").Experiment
    rm = ns.metrics
    rs = np.random.RandomState(12)
    keys = rs.permutation(300 * 120)[:6000]
    data = [("u%d" % (k // 120), "i%d" % (k % 120), float(rs.randint(1, 6))) for k in keys]
    split_kw = dict(test_size=0.2, rating_threshold=4.0, seed=123)
    mf_kw = dict(k=10, max_iter=25, learning_rate=0.01, lambda_reg=0.02, use_bias=True, seed=123)
    bpr_kw = dict(k=10, max_iter=60, learning_rate=0.01, lambda_reg=0.01, seed=123)
    ref = RefExperiment(ns.eval_methods.RatioSplit(data, **split_kw), [ns.MF(**mf_kw), ns.BPR(**bpr_kw)],
                        [rm.MAE(), rm.RMSE(), rm.Recall(k=20), rm.Precision(k=20), rm.AUC(), rm.MAP()], user_based=True)
    ref.run()
    mine = Experiment(RatioSplit(data, **split_kw), [MF(**mf_kw), BPR(**bpr_kw)],
                      [mm.MAE(), mm.RMSE(), mm.Recall(k=20), mm.Precision(k=20), mm.AUC(), mm.MAP()], user_based=True).run()
    for r, m in zip(ref.result, mine.result):
        assert r.model_name == m.model_name and list(r.metric_avg_results) == list(m.metric_avg_results)
        for name, v in r.metric_avg_results.items():
            if "(s)" not in name:   # rank-based values move by one swap when two scores differ in the last ulp
                assert m.metric_avg_results[name] == pytest.approx(v, rel=2e-3, abs=2e-4), (r.model_name, name)
    for ref_model, my_model in zip(ref.models, mine.models):
        assert_close(my_model.u_factors, ref_model.u_factors)
        assert_close(my_model.i_factors, ref_model.i_factors)


@pytest.mark.parametrize("name", ["vebpr_small", "vebpr_odd"])
def test_vebpr_host_class_reproduces_the_reference_goldens(device_double, name):
    from cornac_amd import VEBPR, PurchaseViewDataset

    fx = load_golden(name)
    ds = PurchaseViewDataset.build([(int(a), int(b), 1.0) for a, b in zip(fx["pu"], fx["pi"])],
                                   [(int(a), int(b), 1.0) for a, b in zip(fx["vu"], fx["vi"])], seed=1)
    m = VEBPR(k=int(fx["k"]), max_iter=int(fx["epochs"]), learning_rate=float(fx["lr"]), lambda_reg=float(fx["reg"]),
              alpha=float(fx["alpha"]), seed=int(fx["seed"])).fit(ds)
    assert_close(m.u_factor, fx["U"])
    assert_close(m.i_factor, fx["V"])
    assert np.allclose(m.score(1), m.i_factor @ m.u_factor[1], atol=1e-6)
    with pytest.raises(ValueError):
        VEBPR(k=4, seed=1).fit(golden_dataset(load_golden("tiny")))   # needs the view matrix


def test_vebpr_float64_host_class_reproduces_the_references_float64_run(device_double):
    """float64 init_params through cornac_amd.VEBPR (device double = the oracle's double loop): the reference's own
    float64 result; the tables stay the caller's float64 arrays; score(user, item) works and score(user) fails exactly as
    the reference's does (it allocates a float32 output for float64 tables, recom_vebpr.pyx:356-357); a float32 / float64
    mix is refused before anything is trained"""
    from cornac_amd import VEBPR, PurchaseViewDataset

    fx = load_golden("vebpr_f64")
    ds = PurchaseViewDataset.build([(int(a), int(b), 1.0) for a, b in zip(fx["pu"], fx["pi"])],
                                   [(int(a), int(b), 1.0) for a, b in zip(fx["vu"], fx["vi"])], seed=1)
    kw = dict(k=int(fx["k"]), max_iter=int(fx["epochs"]), learning_rate=float(fx["lr"]), lambda_reg=float(fx["reg"]),
              alpha=float(fx["alpha"]), seed=int(fx["seed"]))
    ip = {"U": fx["init_U"].copy(), "V": fx["init_V"].copy()}
    m = VEBPR(init_params=ip, **kw).fit(ds)
    assert m.u_factor is ip["U"] and m.u_factor.dtype == np.float64
    assert np.abs(m.u_factor - fx["U"]).max() <= 1e-13 and np.abs(m.i_factor - fx["V"]).max() <= 1e-13
    assert abs(m.score(0, 3) - float(fx["score_0_3"])) <= 1e-13
    with pytest.raises(ValueError, match="Buffer dtype mismatch"):
        m.score(0)
    with pytest.raises(ValueError, match="Buffer dtype mismatch"):
        VEBPR(init_params={"U": fx["init_U"].copy(), "V": fx["init_V"].astype(np.float32)}, **kw).fit(ds)


DROPOUT_CASES = [("sgd", True, 0.3), ("adam", True, 0.5), ("rmsprop", False, 0.2), ("adagrad", True, 0.1)]


@pytest.mark.parametrize("opt,use_bias,p", DROPOUT_CASES)
def test_mf_minibatch_dropout_host_class_reproduces_the_reference_goldens(device_double, opt, use_bias, p):
    """MF(backend="pytorch", dropout=p) of the real reference (goldens made by tests/golden/make_golden.py) against the
    host class over the device double: the keep masks cornac_amd/mf.py draws from torch's CPU generator are the
    reference's (seed, the embedding initialisations that consume the generator first, user rows then item rows of
    every batch), and the step applies them as nn.Dropout does"""
    from cornac_amd import MF

    fx = load_golden("mf_minibatch_dropout")
    m = MF(k=int(fx["k"]), backend="hip-minibatch", optimizer=opt, max_iter=int(fx["epochs"]),
           batch_size=int(fx["batch_size"]), learning_rate=float(fx["lr"]), lambda_reg=float(fx["reg"]),
           use_bias=use_bias, dropout=p, seed=int(fx["seed"])).fit(golden_dataset(fx))
    tag = "%s_p%d%s" % (opt, round(100 * p), "" if use_bias else "_nobias")
    for got, key in ((m.u_factors, "_U"), (m.i_factors, "_V"), (m.u_biases, "_Bu"), (m.i_biases, "_Bi")):
        # (Adam divides by sqrt(v): a last-bit difference of a nearly dropped-out row's tiny gradient is amplified)
        assert np.abs(np.asarray(got) - fx[tag + key]).max() <= 2e-5, tag + key
    with pytest.raises(ValueError, match="dropout probability"):
        MF(k=4, backend="hip-minibatch", dropout=1.5).fit(golden_dataset(fx))


@pytest.mark.parametrize("use_bias", [True, False])
@pytest.mark.parametrize("opt", ["sgd", "adam", "rmsprop", "adagrad"])
def test_mf_minibatch_host_class_reproduces_the_reference_goldens(device_double, opt, use_bias):
    from cornac_amd import MF

    fx = load_golden("mf_minibatch")
    m = MF(k=int(fx["k"]), backend="hip-minibatch", optimizer=opt, max_iter=int(fx["epochs"]),
           batch_size=int(fx["batch_size"]), learning_rate=float(fx["lr"]), lambda_reg=float(fx["reg"]),
           use_bias=use_bias, seed=int(fx["seed"])).fit(golden_dataset(fx))
    tag = opt + ("" if use_bias else "_nobias")
    for got, key in ((m.u_factors, "_U"), (m.i_factors, "_V"), (m.u_biases, "_Bu"), (m.i_biases, "_Bi")):
        assert np.abs(np.asarray(got) - fx[tag + key]).max() <= 2e-6, tag + key
    assert len(m.loss_history) == int(fx["epochs"]) and m.loss_history[-1] < m.loss_history[0]
    with pytest.raises(KeyError):
        MF(k=4, backend="hip-minibatch", optimizer="lbfgs").fit(golden_dataset(fx))
    with pytest.raises(ValueError):
        MF(k=4, backend="tensorflow").fit(golden_dataset(fx))


def test_wmf_host_class_drives_the_batches_of_the_fixture(device_double):
    from cornac_amd import WMF, Dataset

    fx = load_golden("wmf_small")
    ds = Dataset.from_uir([(int(u), int(i), float(r)) for u, i, r in zip(fx["users"], fx["items"], fx["ratings"])], seed=123)
    kw = dict(k=int(fx["k"]), lambda_u=float(fx["lambda_u"]), lambda_v=float(fx["lambda_v"]), a=float(fx["a"]),
              b=float(fx["b"]), learning_rate=float(fx["lr"]), batch_size=int(fx["batch_size"]), max_iter=int(fx["max_iter"]))
    m = WMF(verbose=False, seed=7, init_params={"U": fx["U0"].copy(), "V": fx["V0"].copy()}, **kw).fit(ds)
    assert np.abs(m.U - fx["U"]).max() <= 2e-6 and np.abs(m.V - fx["V"]).max() <= 2e-6   # same init, same item batches
    assert m.loss_history[-1] < m.loss_history[0] and len(m.loss_history) == kw["max_iter"]
    with pytest.raises(ValueError):
        WMF(k=4, batch_size=500, verbose=False).fit(ds)


def test_vbpr_host_class_reproduces_the_reference_golden(device_double):
    """initialisation order, the mirrored `uij_iter` sampler (dataset generator re-seeded by fit) and the parameter
    hand-over of cornac_amd.VBPR: with the torch body of the reference behind the double, the real reference's learned
    tables come out"""
    from cornac_amd import VBPR
    from test_oracle_golden import _vbpr_case

    fx, ds, kw = _vbpr_case()
    m = VBPR(verbose=False, **kw).fit(ds)
    for name, key in (("beta_item", "Bi"), ("gamma_user", "Gu"), ("gamma_item", "Gi"), ("theta_user", "Tu"),
                      ("emb_matrix", "E"), ("beta_prime", "Bp"), ("theta_item", "theta_item"),
                      ("visual_bias", "visual_bias")):
        assert np.abs(np.asarray(getattr(m, name)).reshape(fx[key].shape) - fx[key]).max() < 1e-6, name
    s = m.score(3)
    want = m.beta_item + m.visual_bias + m.gamma_item @ m.gamma_user[3] + m.theta_item @ m.theta_user[3]
    assert np.abs(s - want).max() < 1e-5 and len(m.loss_history) == kw["n_epochs"]
    from cornac_amd.recommender import CornacException

    ds.item_image = None
    with pytest.raises(CornacException):
        VBPR(verbose=False, **kw).fit(ds)


@pytest.mark.parametrize("n_ratings,split_kw", [
    (4000, dict(test_size=0.2, val_size=0.1, rating_threshold=3.0, seed=5, exclude_unknowns=False)),
    (4000, dict(test_size=0.3, rating_threshold=1.0, seed=9, exclude_unknowns=True)),
    (500, dict(test_size=0.3, val_size=0.1, rating_threshold=3.0, seed=5, exclude_unknowns=False))])   # sparse: test-only items
def test_experiment_reports_equal_the_reference_reports(device_double, n_ratings, split_kw):
    """the reference's Experiment over MF / BPR / WBPR with ten metrics (rating, @k, full-list, mixed in one list — the
    case where the reference asks rank() for max_k items but its metrics over the whole list read past them), test and
    validation tables, unknown users / items kept or dropped, against cornac_amd's with the device double"""
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ns = ref_loader.load()
    import importlib

    import cornac_amd
    from cornac_amd import Experiment, RatioSplit
    from cornac_amd import metrics as mm

    RefExperiment = importlib.import_module("cornac.experiment").Experiment
    rs = np.random.RandomState(3)
    keys = rs.permutation(250 * 100)[:n_ratings]
    data = [("u%d" % (k // 100), "i%d" % (k % 100), float(rs.randint(1, 6))) for k in keys]

    def models(N):
        return [N.MF(k=8, max_iter=15, learning_rate=0.01, lambda_reg=0.02, use_bias=True, seed=1),
                N.BPR(k=8, max_iter=30, learning_rate=0.01, lambda_reg=0.01, seed=2),
                N.WBPR(k=8, max_iter=30, learning_rate=0.01, lambda_reg=0.01, seed=3)]

    def metrics(M):
        return [M.MAE(), M.RMSE(), M.Recall(k=10), M.NDCG(k=10), M.NCRR(k=5), M.AUC(), M.MAP(), M.MRR(), M.HitRatio(k=3),
                M.FMeasure(k=5)]

    ref = RefExperiment(ns.eval_methods.RatioSplit(data, **split_kw), models(ns), metrics(ns.metrics), user_based=True)
    ref.run()
    mine = Experiment(RatioSplit(data, **split_kw), models(cornac_amd), metrics(mm), user_based=True).run()
    compared = 0
    for which in ("result", "val_result"):
        for r, m in zip(getattr(ref, which) or [], getattr(mine, which) or []):
            assert r.model_name == m.model_name and list(r.metric_avg_results) == list(m.metric_avg_results)
            for name, v in r.metric_avg_results.items():
                if "(s)" not in name:
                    assert m.metric_avg_results[name] == pytest.approx(v, rel=2e-3, abs=2e-4), (which, r.model_name, name)
                    compared += 1
    assert compared >= 30


def _same_rows(R, M, min_rows=1):
    n = 0
    for r, m in zip(R, M):
        assert list(r.metric_avg_results) == list(m.metric_avg_results)
        for name, v in r.metric_avg_results.items():
            if "(s)" not in name:
                assert m.metric_avg_results[name] == pytest.approx(v, rel=2e-3, abs=2e-4), (r.model_name, name)
                n += 1
    assert n >= min_rows


def test_search_cross_validation_and_modalities_equal_the_reference(device_double):
    """three more caller flows against the live reference: GridSearch around BPR (same best point and score), 3-fold
    CrossValidation of MF (same per-fold rows), VBPR fed by an ImageModality that the evaluation method builds"""
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ns = ref_loader.load()
    import importlib

    from cornac_amd import BPR, MF, VBPR, CrossValidation, ImageModality, RatioSplit
    from cornac_amd import hyperopt as my_h
    from cornac_amd import metrics as mm

    ref_h, rm = importlib.import_module("cornac.hyperopt"), ns.metrics
    rs = np.random.RandomState(3)
    keys = rs.permutation(200 * 80)[:3000]
    data = [("u%d" % (k // 80), "i%d" % (k % 80), float(rs.randint(1, 6))) for k in keys]
    skw = dict(test_size=0.2, val_size=0.15, rating_threshold=3.0, seed=4)
    rmeth, mmeth = ns.eval_methods.RatioSplit(data, **skw), RatioSplit(data, **skw)
    space = lambda h: [h.Discrete("learning_rate", [0.05, 0.005]), h.Discrete("k", [4, 8])]   # noqa: E731
    rgs = ref_h.GridSearch(ns.BPR(k=4, max_iter=20, seed=1), space(ref_h), rm.AUC(), rmeth)
    mgs = my_h.GridSearch(BPR(k=4, max_iter=20, seed=1), space(my_h), mm.AUC(), mmeth)
    rr, _ = rmeth.evaluate(rgs, [rm.AUC(), rm.Recall(k=10)], user_based=True)
    mr, _ = mmeth.evaluate(mgs, [mm.AUC(), mm.Recall(k=10)], user_based=True)
    assert rgs.best_params == mgs.best_params and mgs.best_score == pytest.approx(rgs.best_score, rel=1e-6)
    _same_rows([rr], [mr], 2)

    RefCV = importlib.import_module("cornac.eval_methods").CrossValidation
    r, _ = RefCV(data, n_folds=3, rating_threshold=3.0, seed=6).evaluate(
        ns.MF(k=6, max_iter=10, seed=2), [rm.RMSE(), rm.NDCG(k=10)], user_based=True, show_validation=False)
    m, _ = CrossValidation(data, n_folds=3, rating_threshold=3.0, seed=6).evaluate(
        MF(k=6, max_iter=10, seed=2), [mm.RMSE(), mm.NDCG(k=10)], user_based=True)
    _same_rows(list(r), list(m), 6)

    RefImage = importlib.import_module("cornac.data").ImageModality
    RefVBPR = importlib.import_module("cornac.models.vbpr").VBPR
    item_ids = sorted({t[1] for t in data})
    F = np.random.RandomState(1).uniform(0, 1, (len(item_ids), 12)).astype(np.float32)
    kw = dict(test_size=0.2, rating_threshold=1.0, seed=8, exclude_unknowns=True)
    rsp = ns.eval_methods.RatioSplit(data, item_image=RefImage(features=F.copy(), ids=list(item_ids), normalized=True), **kw)
    msp = RatioSplit(data, item_image=ImageModality(features=F.copy(), ids=list(item_ids), normalized=True), **kw)
    vkw = dict(k=4, k2=4, n_epochs=3, batch_size=64, learning_rate=0.005, lambda_w=0.01, lambda_b=0.01, lambda_e=0.0, seed=3,
               verbose=False)
    ref_model, my_model = RefVBPR(use_gpu=False, **vkw), VBPR(**vkw)
    rres, _ = rsp.evaluate(ref_model, [rm.AUC(), rm.Recall(k=10)], user_based=True)
    mres, _ = msp.evaluate(my_model, [mm.AUC(), mm.Recall(k=10)], user_based=True)
    _same_rows([rres], [mres], 2)
    assert np.abs(my_model.gamma_user - ref_model.gamma_user).max() < 1e-6
    assert np.abs(my_model.theta_item - ref_model.theta_item).max() < 1e-5


@pytest.mark.parametrize("n_ratings", [2500, 260])
def test_recommender_surface_equals_the_reference_recommender(device_double, n_ratings):
    """rank / score / rate / recommend of fitted BPR, MF, WBPR against the live reference's, call by call — known and
    unknown users and items (the sparse case leaves test-only users and items in the global id maps), candidate
    lists, exceptions.  Scores agree to the BLAS-order tolerance; orders are compared where the reference defines them
    (its tie order and the tail behind the first k are unspecified)."""
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ns = ref_loader.load()
    from cornac_amd import BPR, MF, WBPR, RatioSplit

    rs = np.random.RandomState(5)
    keys = rs.permutation(120 * 70)[:n_ratings]
    data = [("u%d" % (k // 70), "i%d" % (k % 70), float(rs.randint(1, 6))) for k in keys]
    kw = dict(test_size=0.4, seed=2, exclude_unknowns=False)
    rsp, msp = ns.eval_methods.RatioSplit(data, **kw), RatioSplit(data, **kw)
    n_train_users, n_train_items = rsp.train_set.num_users, rsp.train_set.num_items
    if n_ratings < 1000:
        assert rsp.total_items > n_train_items and rsp.total_users > n_train_users

    def outcome(fn, *a, **k):
        try:
            return fn(*a, **k)
        except Exception as e:   # noqa: BLE001 - the exception type is what is compared
            return type(e).__name__

    for R, M, mk in ((ns.BPR, BPR, dict(k=6, max_iter=20, seed=1)), (ns.MF, MF, dict(k=6, max_iter=15, seed=1, use_bias=True)),
                     (ns.WBPR, WBPR, dict(k=6, max_iter=10, seed=4))):
        r, m = R(**mk).fit(rsp.train_set), M(**mk).fit(msp.train_set)
        for u in (0, 5, n_train_users - 1, rsp.total_users - 1):
            for cand in (None, np.arange(3, 40), np.array([rsp.total_items - 1, 2, 7, 30])):
                for k in (-1, 3):
                    a, b = outcome(r.rank, u, item_indices=cand, k=k), outcome(m.rank, u, item_indices=cand, k=k)
                    assert isinstance(a, str) == isinstance(b, str) and (not isinstance(a, str) or a == b)
                    if isinstance(a, str):
                        continue
                    assert np.allclose(a[1], b[1], atol=3e-5)
                    n = len(a[1]) if k == -1 else k
                    gaps = np.abs(np.diff(np.sort(a[1]))) if len(a[1]) > 1 else np.array([1.0])
                    if gaps.min() > 1e-4:            # no (near-)ties: the order is defined
                        assert list(a[0][:n]) == list(b[0][:n]), (R.__name__, u, k)
            for it in (0, 3, n_train_items - 1, rsp.total_items - 1, rsp.total_items + 5):
                for uu in (u, rsp.total_users + 3):
                    for fn in ("score", "rate"):
                        a, b = outcome(getattr(r, fn), uu, it), outcome(getattr(m, fn), uu, it)
                        if isinstance(a, str) or isinstance(b, str):
                            assert a == b, (R.__name__, fn, uu, it, a, b)
                        else:
                            assert abs(float(a) - float(b)) <= 3e-5, (R.__name__, fn, uu, it)
        uid = rsp.train_set.user_ids[5]
        assert list(r.recommend(uid, k=5)) == list(m.recommend(uid, k=5))
        unseen_ref = r.recommend(uid, k=7, remove_seen=True, train_set=rsp.train_set)
        assert list(unseen_ref) == list(m.recommend(uid, k=7, remove_seen=True, train_set=msp.train_set))
        for args in (("nobody",), (uid, 10 ** 6)):
            assert outcome(r.recommend, *args) == outcome(m.recommend, *args)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_purchase_view_dataset_and_vebpr_equal_the_live_reference(device_double, seed):
    """views with users and items the purchases never mention, duplicated view records and views of purchased pairs:
    the same purchase / view matrices as the reference's PurchaseViewDataset, and through cornac_amd.VEBPR (device
    double) the same factors, scores and ranking as the reference's compiled VEBPR"""
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ref_loader.load()
    import importlib
    import warnings

    from cornac_amd import VEBPR, PurchaseViewDataset

    RefPV = importlib.import_module("cornac.data").PurchaseViewDataset
    RefVEBPR = importlib.import_module("cornac.models.bpr.recom_vebpr").VEBPR
    rs = np.random.RandomState(seed)

    def pairs(nu, ni, n, dup=False):
        keys = rs.randint(0, nu * ni, n) if dup else rs.permutation(nu * ni)[:n]
        return [("u%d" % (k // ni), "i%d" % (k % ni), 1.0) for k in keys]

    purchase = pairs(40, 30, 300)
    views = pairs(55, 36, 500, dup=True) + purchase[:40]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r, m = RefPV.build(purchase, views, seed=4), PurchaseViewDataset.build(purchase, views, seed=4)
    assert (r.matrix != m.matrix).nnz == 0 and r.view_matrix.shape == m.view_matrix.shape
    assert (r.view_matrix != m.view_matrix).nnz == 0 and (r.num_users, r.num_items) == (m.num_users, m.num_items)
    kw = dict(k=5, max_iter=8, learning_rate=0.05, lambda_reg=0.01, alpha=0.4, seed=7)
    a, b = RefVEBPR(**kw).fit(r), VEBPR(**kw).fit(m)
    assert_close(b.u_factor, a.u_factor)
    assert_close(b.i_factor, a.i_factor)
    assert np.abs(a.score(3) - b.score(3)).max() < 1e-6
    assert list(a.rank(3, k=5)[0][:5]) == list(b.rank(3, k=5)[0][:5])
