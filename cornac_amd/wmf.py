"""WMF on MI355X — constructor, learned attributes (`U`, `V`) and `fit/score/rank` surface of the
reference's `cornac.models.WMF` (cornac/models/wmf/recom_wmf.py:25-270).  The epoch loop keeps the
reference's host iterator (`train_set.item_iter(batch_size, shuffle=True)`); the TensorFlow graph
and its Adam step (cornac/models/wmf/wmf.py:34-55) are replaced by `cornac_hip_wmf_fit_batches`."""
import numpy as np

from . import _lib
from .recommender import Recommender, ScoreException
from .vbpr import _xavier_uniform


class WMF(Recommender):
    def __init__(self, name="WMF", k=200, lambda_u=0.01, lambda_v=0.01, a=1, b=0.01, learning_rate=0.001,
                 batch_size=128, max_iter=100, trainable=True, verbose=True, init_params=None, seed=None, device=0):
        super().__init__(name=name, trainable=trainable, verbose=verbose)
        self.k, self.lambda_u, self.lambda_v, self.a, self.b = k, lambda_u, lambda_v, a, b
        self.learning_rate, self.batch_size, self.max_iter = learning_rate, batch_size, max_iter
        self.seed = seed
        self.device = device
        self.init_params = {} if init_params is None else init_params
        self.U = self.init_params.get("U", None)
        self.V = self.init_params.get("V", None)

    def _init(self):
        rng = np.random.RandomState(self.seed)  # recom_wmf.py:121-126 (get_rng)
        if self.U is None:
            self.U = _xavier_uniform((self.num_users, self.k), rng)
        if self.V is None:
            self.V = _xavier_uniform((self.num_items, self.k), rng)

    def fit(self, train_set, val_set=None):
        Recommender.fit(self, train_set, val_set)
        self._init()
        if self.trainable:
            self._fit_cf(train_set)
        self._drop_scorer()
        return self

    def _fit_cf(self, train_set):
        if not 1 <= self.batch_size <= 128:
            raise ValueError("the HIP backend supports 1 <= batch_size <= 128 (got %d)" % self.batch_size)
        trainer = _lib.WmfTrainer(train_set.csc_matrix, self.k, device=self.device)
        try:
            trainer.set_factors(self.U, self.V)
            self.loss_history = []
            for _ in range(self.max_iter):
                # recom_wmf.py:181-199: one optimiser step per shuffled batch of item ids
                batches = list(train_set.item_iter(self.batch_size, shuffle=True))
                losses = trainer.fit_batches(batches, self.lambda_u, self.lambda_v, self.a, self.b,
                                             self.learning_rate)
                self.loss_history.append(float(np.sum(losses)) / sum(len(x) for x in batches))
            self.U, self.V = trainer.get_factors()
        finally:
            trainer.close()
        if self.verbose:
            print("Learning completed!")

    def _scoring_tables(self):
        return self.U, self.V, None, None

    def score(self, user_idx, item_idx=None):
        """recom_wmf.py:214-240"""
        if self.is_unknown_user(user_idx):
            raise ScoreException("Can't make score prediction for user %d" % user_idx)
        if item_idx is not None and self.is_unknown_item(item_idx):
            raise ScoreException("Can't make score prediction for item %d" % item_idx)
        if item_idx is None:
            return self._get_scorer().score_user(user_idx)
        return self.V[item_idx, :].dot(self.U[user_idx, :])

    def get_vector_measure(self):
        return "dot"

    def get_user_vectors(self):
        return self.U

    def get_item_vectors(self):
        return self.V
