"""Hyper-parameter search over a model of this path — `GridSearch` / `RandomSearch` with `Discrete` / `Continuous`
domains, the reference's interface and enumeration order (cornac/hyperopt.py:28-289): domains sorted by name, grid
points as the product of each domain's sorted values, random points drawn domain by domain from one generator seeded
with the model's seed; every point trains a `clone(params)` on the training set and is scored on the VALIDATION set
with the batched evaluation loops; the searcher then behaves as the best model (`score`, `rank`, and the batched
entry points the evaluation loops look for)."""
from itertools import product

import numpy as np

from . import eval as _eval
from .experiment import _rng
from .recommender import Recommender


class SearchDomain:
    def __init__(self, name):
        self.name = name

    def _sample(self, rng):
        raise NotImplementedError()


class Discrete(SearchDomain):
    """a finite list of values"""

    def __init__(self, name, values):
        super().__init__(name)
        self.values = values

    def _sample(self, rng):
        return rng.choice(self.values)


class Continuous(SearchDomain):
    """uniform over [low, high)"""

    def __init__(self, name, low=0.0, high=1.0):
        super().__init__(name)
        self.low, self.high = low, high

    def _sample(self, rng):
        return rng.uniform(low=self.low, high=self.high)


class BaseSearch(Recommender):
    def __init__(self, model, space, metric, eval_method, name="BaseSearch"):
        super().__init__(name=name, verbose=model.verbose)
        self.model, self.metric, self.eval_method = model, metric, eval_method
        self.space = sorted(space, key=lambda d: d.name)
        self.best_score, self.best_model, self.best_params = None, None, None

    def _build_param_set(self):
        raise NotImplementedError()

    def _score(self, model, train_set, val_set):
        if getattr(self.metric, "type", None) == "rating":
            return _eval.rating_eval(model, [self.metric], val_set)[0][0]
        return _eval.ranking_eval(model, [self.metric], train_set, val_set,
                                  rating_threshold=self.eval_method.rating_threshold,
                                  exclude_unknowns=self.eval_method.exclude_unknowns, verbose=False)[0][0]

    def fit(self, train_set, val_set=None):
        assert val_set is not None
        Recommender.fit(self, train_set, val_set)
        higher_better = getattr(self.metric, "higher_better", True)
        self.best_score = -np.inf if higher_better else np.inf
        self.best_model = self.best_params = None
        for params in self._build_param_set():
            if self.verbose:
                print("Evaluating: {}".format(params))
            model = self.model.clone(params).fit(train_set, val_set)
            score = self._score(model, train_set, val_set)
            if (score > self.best_score) if higher_better else (score < self.best_score):
                self.best_score, self.best_model, self.best_params = score, model, params
        if self.verbose:
            print("Best parameter settings: {}".format(self.best_params))
            print("{} = {:.4f}".format(self.metric.name, self.best_score))
        return self

    # the searcher stands in for its best model
    def transform(self, test_set):
        return self.best_model.transform(test_set)

    def score(self, user_idx, *args, **kwargs):
        return self.best_model.score(user_idx, *args, **kwargs)

    def rank(self, user_idx, item_indices=None, k=-1, **kwargs):
        return self.best_model.rank(user_idx, item_indices, k, **kwargs)

    def rate(self, user_idx, item_idx, clipping=True):
        return self.best_model.rate(user_idx, item_idx, clipping)

    # batched entry points exist on the searcher exactly when the best model has them (the evaluation loops probe
    # with hasattr): an AttributeError from the best model propagates through these properties
    @property
    def batch_num_items(self):
        return self.best_model.batch_num_items

    @property
    def rank_batch(self):
        return self.best_model.rank_batch

    @property
    def rate_batch(self):
        return self.best_model.rate_batch

    @property
    def rank_positions_batch(self):
        return self.best_model.rank_positions_batch


class GridSearch(BaseSearch):
    def __init__(self, model, space, metric, eval_method):
        for domain in space:
            if not isinstance(domain, Discrete):
                raise ValueError("GridSearch only supports Discrete domain but {} is not!\n"
                                 "Please consider using RandomSearch instead.".format(domain.name))
        super().__init__(model, space, metric, eval_method, name="GridSearch_{}".format(model.name))

    def _build_param_set(self):
        names = [d.name for d in self.space]
        return [dict(zip(names, point)) for point in product(*(sorted(d.values) for d in self.space))]


class RandomSearch(BaseSearch):
    def __init__(self, model, space, metric, eval_method, n_trails=10):
        super().__init__(model, space, metric, eval_method, name="RandomSearch_{}".format(model.name))
        self.n_trails = n_trails

    def _build_param_set(self):
        names = [d.name for d in self.space]
        rng = _rng(getattr(self.model, "seed", None))
        return [dict(zip(names, [d._sample(rng) for d in self.space])) for _ in range(self.n_trails)]
