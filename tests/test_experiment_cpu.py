"""cornac_amd.experiment (RatioSplit / BaseMethod / Experiment): the reference's own expectations
(tests/cornac/eval_methods/test_ratio_split.py:31-104, test_base_method.py) and a live comparison with the reference's
classes where /root/reference is present.  Host stand-in models only — no GPU."""
import itertools
import os
import random

import numpy as np
import pytest

from cornac_amd import BaseMethod, CrossValidation, Experiment, RatioSplit, StratifiedSplit
from cornac_amd import metrics as mm


class TableModel:
    """scores from a fixed random table over the global id space; same tie rule as the device kernels"""
    name = "table"

    def __init__(self, seed=0):
        self.seed = seed

    def fit(self, train_set, val_set=None):
        self.num_users, self.num_items = train_set.num_users, train_set.num_items
        if not isinstance(getattr(type(self), "total_items", None), property):   # a property on the reference's base class
            self.total_items = len(train_set.iid_map)
        self.min_rating, self.max_rating = train_set.min_rating, train_set.max_rating
        self.S = np.random.RandomState(self.seed).uniform(0.5, 5.5, (train_set.num_users, train_set.num_items))
        return self

    def transform(self, test_set):
        pass

    def clone(self, new_params=None):
        return TableModel(self.seed)

    def score(self, user_idx, item_idx=None):
        return self.S[user_idx] if item_idx is None else self.S[user_idx, item_idx]

    def rate(self, user_idx, item_idx, clipping=True):
        if user_idx >= self.num_users or item_idx >= self.num_items:
            return np.float64(3.0)
        s = self.S[user_idx, item_idx]
        return np.clip(s, self.min_rating, self.max_rating) if clipping else s

    def rank(self, user_idx, item_indices=None, k=-1, **kw):
        sc = self.S[user_idx]
        item_indices = np.arange(self.num_items) if item_indices is None else np.asarray(item_indices)
        sc = sc[item_indices]
        return item_indices[np.argsort(sc, kind="stable")[::-1]], sc   # every candidate, like Recommender.rank


def _grid():
    rnd = random.Random(5)
    return [(u, i, rnd.randint(1, 5)) for u, i in itertools.product(["u%d" % a for a in range(12)],
                                                                     ["i%d" % b for b in range(9)])]


def test_validate_size_values_of_the_reference_test():
    assert RatioSplit.validate_size(0.1, 0.2, 10) == (7, 1, 2)
    assert RatioSplit.validate_size(None, 0.5, 10) == (5, 0, 5)
    assert RatioSplit.validate_size(None, None, 10) == (10, 0, 0)
    assert RatioSplit.validate_size(2, 2, 10) == (6, 2, 2)
    for bad in ((-1, 0.2), (1, -0.2), (11, 0.2), (0, 11), (3, 8)):
        with pytest.raises(ValueError):
            RatioSplit.validate_size(bad[0], bad[1], 10)


def test_split_sizes_and_reproducibility():
    data = [(u, i, random.randint(1, 5)) for u, i in itertools.product(["u1", "u2", "u3", "u4"],
                                                                        ["i1", "i2", "i3", "i4", "i5"])]
    rs = RatioSplit(data, test_size=0.1, val_size=0.1, seed=123)
    assert (rs.train_size, rs.test_size, rs.val_size) == (16, 2, 2)
    again = RatioSplit(data, test_size=0.1, val_size=0.1, seed=123)
    for a, b in zip(rs.train_set.uir_tuple, again.train_set.uir_tuple):
        assert np.array_equal(a, b)
    assert rs.train_set.num_ratings + rs.test_set.num_ratings + rs.val_set.num_ratings <= 20
    with pytest.raises(ValueError):          # nothing left of a held-out part once unknown users/items are dropped
        RatioSplit([("a", "x", 1.0), ("b", "y", 2.0), ("c", "z", 3.0)], test_size=1, seed=1)


def test_evaluate_and_experiment_run(capsys, tmp_path):
    method = RatioSplit(_grid(), test_size=0.2, val_size=0.1, rating_threshold=3.0, seed=7, exclude_unknowns=True)
    metrics = [mm.MAE(), mm.RMSE(), mm.Recall(k=[3, 5]), mm.NDCG(k=-1), mm.AUC()]
    test_res, val_res = method.evaluate(TableModel(), metrics, user_based=True)
    keys = list(test_res.metric_avg_results)
    assert keys == ["MAE", "RMSE", "AUC", "NDCG@-1", "Recall@3", "Recall@5", "Train (s)", "Test (s)"]
    assert list(val_res.metric_avg_results)[-1] == "Time (s)"
    assert 0 < test_res.metric_avg_results["AUC"] < 1 and len(test_res.metric_user_results["Recall@3"]) > 0
    exp = Experiment(method, [TableModel(0), TableModel(1)], metrics, user_based=False).run()
    out = capsys.readouterr().out
    assert "TEST:" in out and "VALIDATION:" in out and out.count("table") == 4
    assert len(exp.result) == 2 and exp.result[0].metric_avg_results["MAE"] != exp.result[1].metric_avg_results["MAE"]
    class Saveable(TableModel):
        def save(self, save_dir):
            open(os.path.join(save_dir, "saved_%d" % self.seed), "w").close()

    Experiment(method, [Saveable(5)], [mm.MAE()], save_dir=str(tmp_path / "out")).run()
    written = sorted(os.listdir(str(tmp_path / "out")))
    assert written[0].startswith("CornacExp-") and written[0].endswith(".log") and written[1] == "saved_5"
    fs = BaseMethod.from_splits(_grid()[:80], _grid()[80:], rating_threshold=3.0, exclude_unknowns=True, seed=3)
    res, none = fs.evaluate(TableModel(), [mm.Precision(k=2)], user_based=True)
    assert none is None and "Precision@2" in res.metric_avg_results


def test_split_and_evaluation_match_the_reference_classes():
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ns = ref_loader.load()
    rm, RefSplit = ns.metrics, ns.eval_methods.RatioSplit
    rs = np.random.RandomState(2)
    keys = rs.permutation(60 * 40)[:900]
    data = [("u%d" % (k // 40), "i%d" % (k % 40), float(rs.randint(1, 6))) for k in keys]
    for kw in (dict(test_size=0.2, seed=11), dict(test_size=0.15, val_size=0.1, seed=5, exclude_unknowns=False),
               dict(test_size=100, val_size=50, seed=9, rating_threshold=4.0)):
        ref, mine = RefSplit(data, **kw), RatioSplit(data, **kw)
        for part in ("train_set", "test_set", "val_set"):
            a, b = getattr(ref, part), getattr(mine, part)
            assert (a is None) == (b is None)
            if a is None:
                continue
            assert (a.num_users, a.num_items) == (b.num_users, b.num_items)
            assert list(a.uid_map.items()) == list(b.uid_map.items()) and list(a.iid_map.items()) == list(b.iid_map.items())
            for x, y in zip(a.uir_tuple, b.uir_tuple):
                assert np.array_equal(x, y)
        assert (ref.total_users, ref.total_items) == (mine.total_users, mine.total_items)
        for user_based in (True, False):
            ref_metrics = [rm.MAE(), rm.RMSE(), rm.Recall(k=[3, 10]), rm.NDCG(k=5), rm.AUC(), rm.MAP()]
            my_metrics = [mm.MAE(), mm.RMSE(), mm.Recall(k=[3, 10]), mm.NDCG(k=5), mm.AUC(), mm.MAP()]
            r_test, r_val = ref.evaluate(TableModel(3), ref_metrics, user_based=user_based)
            m_test, m_val = mine.evaluate(TableModel(3), my_metrics, user_based=user_based)
            for r, m in ((r_test, m_test), (r_val, m_val)):
                assert (r is None) == (m is None)
                if r is None:
                    continue
                assert list(r.metric_avg_results) == list(m.metric_avg_results)
                for name, v in r.metric_avg_results.items():
                    if "(s)" not in name:
                        assert m.metric_avg_results[name] == pytest.approx(v, rel=1e-9), (kw, name)
                        assert r.metric_user_results[name].keys() == m.metric_user_results[name].keys()


def _uirt(n_users=30, n_items=25, n=500, seed=4):
    rs = np.random.RandomState(seed)
    keys = rs.permutation(n_users * n_items)[:n]
    return [("u%d" % (k // n_items), "i%d" % (k % n_items), float(rs.randint(1, 6)), int(rs.randint(0, 10000))) for k in keys]


def test_cross_validation_and_stratified_split_on_their_own(capsys):
    data = _grid()
    cv = CrossValidation(data, n_folds=4, rating_threshold=3.0, seed=2)
    assert np.bincount(cv._partition).tolist() == [27, 27, 27, 27]
    res, none = cv.evaluate(TableModel(), [mm.MAE(), mm.Recall(k=3)], user_based=True)
    assert none is None and len(res) == 4 and set(res.metric_mean) == {"MAE", "Recall@3", "Train (s)", "Test (s)"}
    assert res.metric_mean["MAE"] == pytest.approx(np.mean([r.metric_avg_results["MAE"] for r in res]))
    assert "Fold 3" in str(res) and "Mean" in str(res) and "Std" in str(res)
    with pytest.raises(ValueError):
        CrossValidation(data, n_folds=4, partition=[0, 1, 2])
    with pytest.raises(ValueError):
        CrossValidation(data, n_folds=4, partition=np.zeros(len(data), int))
    Experiment(CrossValidation(data, n_folds=3, seed=1), [TableModel()], [mm.RMSE()]).run()
    assert "Fold 2" in capsys.readouterr().out
    uirt = _uirt()
    st = StratifiedSplit(uirt, group_by="user", chrono=True, test_size=0.25, seed=6, exclude_unknowns=False)
    newest_train = {}
    for u, t in zip(st.train_set.uir_tuple[0], st.train_set.timestamps):
        newest_train[u] = max(newest_train.get(u, -1), t)
    for u, t in zip(st.test_set.uir_tuple[0], st.test_set.timestamps):     # held-out ratings are the user's newest
        assert t >= newest_train[u]
    with pytest.raises(ValueError):
        StratifiedSplit(uirt, group_by="basket")
    with pytest.raises(ValueError):
        StratifiedSplit(_grid(), chrono=True, fmt="UIR")


def test_cross_validation_and_stratified_split_match_the_reference_classes():
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ns = ref_loader.load()
    import importlib

    RefCV = importlib.import_module("cornac.eval_methods").CrossValidation
    RefStrat = importlib.import_module("cornac.eval_methods").StratifiedSplit
    rm = ns.metrics
    data = _grid()
    for n_folds, seed in ((5, 3), (4, 8), (7, 1)):
        ref, mine = RefCV(data, n_folds=n_folds, rating_threshold=3.0, seed=seed), \
            CrossValidation(data, n_folds=n_folds, rating_threshold=3.0, seed=seed)
        assert np.array_equal(ref._partition, mine._partition)
        r, _ = ref.evaluate(TableModel(2), [rm.MAE(), rm.Recall(k=3), rm.AUC()], user_based=True, show_validation=False)
        m, _ = mine.evaluate(TableModel(2), [mm.MAE(), mm.Recall(k=3), mm.AUC()], user_based=True)
        assert len(r) == len(m) == n_folds
        for name in ("MAE", "Recall@3", "AUC"):
            assert m.metric_mean[name] == pytest.approx(r.metric_mean[name], rel=1e-9)
            assert m.metric_std[name] == pytest.approx(r.metric_std[name], rel=1e-9, abs=1e-15)
            for fr, fm in zip(r, m):
                assert fm.metric_avg_results[name] == pytest.approx(fr.metric_avg_results[name], rel=1e-9)
    uirt = _uirt()
    for kw in (dict(group_by="user", chrono=True, test_size=0.2, seed=3), dict(group_by="item", chrono=False, test_size=0.3, seed=5),
               dict(group_by="user", chrono=True, test_size=0.2, val_size=0.1, seed=7, exclude_unknowns=False),
               dict(group_by="item", chrono=True, test_size=2, seed=2)):
        ref, mine = RefStrat(uirt, **kw), StratifiedSplit(uirt, **kw)
        for part in ("train_set", "test_set", "val_set"):
            a, b = getattr(ref, part), getattr(mine, part)
            assert (a is None) == (b is None)
            if a is not None:
                for x, y in zip(a.uir_tuple, b.uir_tuple):
                    assert np.array_equal(x, y)
                assert np.array_equal(a.timestamps, b.timestamps)
                assert list(a.uid_map.items()) == list(b.uid_map.items())


def test_base_method_expectations_of_the_reference_tests():
    # tests/cornac/eval_methods/test_base_method.py:33-84
    bm = BaseMethod(None, verbose=False)
    assert bm.exclude_unknowns and bm.rating_threshold == 1.0
    with pytest.raises(ValueError):
        bm.evaluate(None, {}, False)                    # no training set
    from cornac_amd import Dataset

    data = _grid()[:40]
    bm.train_set = Dataset.from_uir(data)
    with pytest.raises(ValueError):
        bm.evaluate(None, {}, False)                    # no test set
    for train, test in ((None, None), (data, None), (data, [])):
        with pytest.raises(ValueError):
            BaseMethod.from_splits(train_data=train, test_data=test, exclude_unknowns=True)
    n_users, n_items = len({t[0] for t in data}), len({t[1] for t in data})
    bm = BaseMethod.from_splits(train_data=data[:-1], test_data=data[-1:])
    assert (bm.total_users, bm.total_items) == (n_users, n_items) and bm.val_set is None
    bm = BaseMethod.from_splits(train_data=data[:-1], test_data=data[-1:], val_data=[(data[0][0], data[1][1], 5.0)])
    assert (bm.total_users, bm.total_items) == (n_users, n_items) and bm.val_set.num_ratings == 1
    with pytest.raises(ValueError):
        BaseMethod.organize_metrics("MAE")
    rating, ranking = BaseMethod.organize_metrics({"rating": [mm.MAE()], "ranking": [mm.AUC()]})
    assert [m.name for m in rating + ranking] == ["MAE", "AUC"]
