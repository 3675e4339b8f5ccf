"""cornac_amd.hyperopt: enumeration order, selection and delegation; live comparison with the reference's searchers
(cornac/hyperopt.py) where /root/reference is present.  Host stand-in model only."""
import numpy as np
import pytest

from cornac_amd import RatioSplit
from cornac_amd import metrics as mm
from cornac_amd.hyperopt import Continuous, Discrete, GridSearch, RandomSearch
from test_experiment_cpu import TableModel, _grid


class Tunable(TableModel):
    """prediction noise grows with |alpha - 0.3| and with beta: the best grid point is known"""
    verbose = False

    def __init__(self, alpha=1.0, beta=2, seed=4):
        super().__init__(seed)
        self.alpha, self.beta = alpha, beta

    def clone(self, new_params=None):
        p = dict(alpha=self.alpha, beta=self.beta, seed=self.seed)
        p.update(new_params or {})
        return Tunable(**p)

    def fit(self, train_set, val_set=None):
        super().fit(train_set, val_set)
        truth = np.full(self.S.shape, 3.0)
        u, i, r = train_set.uir_tuple
        truth[u, i] = r
        noise = np.random.RandomState(self.seed).normal(size=self.S.shape)
        self.S = truth + (abs(self.alpha - 0.3) + 0.1 * self.beta) * noise
        return self


def _method():
    return RatioSplit(_grid(), test_size=0.2, val_size=0.2, rating_threshold=3.0, seed=3, exclude_unknowns=True)


def test_grid_and_random_search_select_and_delegate():
    method = _method()
    gs = GridSearch(Tunable(), [Discrete("beta", [3, 1, 2]), Discrete("alpha", [0.9, 0.3, 0.6])], mm.RMSE(), method)
    assert gs.name == "GridSearch_table" and [d.name for d in gs.space] == ["alpha", "beta"]
    points = gs._build_param_set()
    assert points[0] == {"alpha": 0.3, "beta": 1} and points[-1] == {"alpha": 0.9, "beta": 3} and len(points) == 9
    gs.fit(method.train_set, method.val_set)
    assert gs.best_params == {"alpha": 0.3, "beta": 1} and gs.best_score < 2.0
    assert np.array_equal(gs.rank(0)[0], gs.best_model.rank(0)[0]) and gs.rate(0, 0) == gs.best_model.rate(0, 0)
    assert not hasattr(gs, "rank_batch") and not hasattr(gs, "rank_positions_batch")   # the stand-in has none
    res, val = method.evaluate(gs, [mm.RMSE(), mm.Recall(k=3)], user_based=False)      # a searcher is a model
    assert "RMSE" in res.metric_avg_results and val is not None
    with pytest.raises(ValueError):
        GridSearch(Tunable(), [Continuous("alpha")], mm.RMSE(), method)
    rnd = RandomSearch(Tunable(seed=11), [Continuous("alpha", 0.0, 1.0), Discrete("beta", [1, 2, 3])], mm.Recall(k=3),
                       method, n_trails=6)
    pts = rnd._build_param_set()
    assert len(pts) == 6 and pts == rnd._build_param_set() and all(0 <= p["alpha"] < 1 and p["beta"] in (1, 2, 3) for p in pts)
    rnd.fit(method.train_set, method.val_set)
    assert rnd.best_params in pts and rnd.best_score == max(
        rnd._score(Tunable(seed=11).clone(p).fit(method.train_set, method.val_set), method.train_set, method.val_set) for p in pts)


def test_search_points_and_selection_match_the_reference_searchers():
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ns = ref_loader.load()
    import importlib

    ref_h = importlib.import_module("cornac.hyperopt")
    rm = ns.metrics

    class RefTunable(Tunable, ns.Recommender):   # the reference searcher insists on its own base class
        def __init__(self, alpha=1.0, beta=2, seed=4):
            ns.Recommender.__init__(self, name="table")
            Tunable.__init__(self, alpha, beta, seed)

        def clone(self, new_params=None):
            p = dict(alpha=self.alpha, beta=self.beta, seed=self.seed)
            p.update(new_params or {})
            return RefTunable(**p)

        def fit(self, train_set, val_set=None):
            ns.Recommender.fit(self, train_set, val_set)
            return Tunable.fit(self, train_set, val_set)

    method = _method()
    ref_method = ns.eval_methods.RatioSplit(_grid(), test_size=0.2, val_size=0.2, rating_threshold=3.0, seed=3,
                                            exclude_unknowns=True)
    space = lambda h: [h.Discrete("beta", [3, 1, 2]), h.Discrete("alpha", [0.9, 0.3, 0.6])]   # noqa: E731
    import cornac_amd.hyperopt as my_h

    for metric_name, kw in (("RMSE", {}), ("Recall", dict(k=3)), ("AUC", {})):
        ref = ref_h.GridSearch(RefTunable(), space(ref_h), getattr(rm, metric_name)(**kw), ref_method)
        mine = GridSearch(Tunable(), space(my_h), getattr(mm, metric_name)(**kw), method)
        assert ref._build_param_set() == mine._build_param_set()
        ref.fit(ref_method.train_set, ref_method.val_set)
        mine.fit(method.train_set, method.val_set)
        assert ref.best_params == mine.best_params and mine.best_score == pytest.approx(ref.best_score, rel=1e-9)
    rspace = lambda h: [h.Continuous("alpha", 0.1, 0.9), h.Discrete("beta", [1, 2, 3])]   # noqa: E731
    ref = ref_h.RandomSearch(RefTunable(seed=21), rspace(ref_h), rm.MAE(), ref_method, n_trails=7)
    mine = RandomSearch(Tunable(seed=21), rspace(my_h), mm.MAE(), method, n_trails=7)
    assert ref._build_param_set() == mine._build_param_set()
    ref.fit(ref_method.train_set, ref_method.val_set)
    mine.fit(method.train_set, method.val_set)
    assert ref.best_params == mine.best_params and mine.best_score == pytest.approx(ref.best_score, rel=1e-9)
