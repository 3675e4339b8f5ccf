"""VBPR on MI355X — constructor, learned attributes (`beta_item, gamma_user, gamma_item, theta_user,
emb_matrix, beta_prime, theta_item, visual_bias`) and `fit/score/rank` surface of the reference's
`cornac.models.VBPR` (cornac/models/vbpr/recom_vbpr.py:30-330).  The minibatch loop keeps the
reference's host sampler (`train_set.uij_iter(batch_size, shuffle=True)`, one epoch at a time) and
replaces the torch forward/backward/Adam body by `cornac_hip_vbpr_fit_batches`."""
import numpy as np

from . import _lib
from .recommender import CornacException, Recommender


def _xavier_uniform(shape, rng):
    # cornac/utils/init_utils.py:116-144
    std = np.sqrt(2.0 / np.sum(shape))
    limit = np.sqrt(3.0) * std
    return rng.uniform(-limit, limit, shape).astype(np.float32)


class VBPR(Recommender):
    def __init__(self, name="VBPR", k=10, k2=10, n_epochs=50, batch_size=100, learning_rate=0.005, lambda_w=0.01,
                 lambda_b=0.01, lambda_e=0.0, use_gpu=True, trainable=True, verbose=True, init_params=None,
                 seed=None, device=0):
        super().__init__(name=name, trainable=trainable, verbose=verbose)
        self.k, self.k2, self.n_epochs, self.batch_size = k, k2, n_epochs, batch_size
        self.learning_rate, self.lambda_w, self.lambda_b, self.lambda_e = learning_rate, lambda_w, lambda_b, lambda_e
        self.use_gpu = use_gpu
        self.seed = seed
        self.device = device
        self.init_params = {} if init_params is None else init_params
        self.beta_item = self.init_params.get("Bi", None)
        self.gamma_user = self.init_params.get("Gu", None)
        self.gamma_item = self.init_params.get("Gi", None)
        self.theta_user = self.init_params.get("Tu", None)
        self.emb_matrix = self.init_params.get("E", None)
        self.beta_prime = self.init_params.get("Bp", None)

    def _init(self, n_users, n_items, features):
        rng = np.random.RandomState(self.seed)  # recom_vbpr.py:117
        self.beta_item = np.zeros(n_items) if self.beta_item is None else self.beta_item
        if self.gamma_user is None:
            self.gamma_user = _xavier_uniform((n_users, self.k), rng)
        if self.gamma_item is None:
            self.gamma_item = _xavier_uniform((n_items, self.k), rng)
        if self.theta_user is None:
            self.theta_user = _xavier_uniform((n_users, self.k2), rng)
        if self.emb_matrix is None:
            self.emb_matrix = _xavier_uniform((features.shape[1], self.k2), rng)
        if self.beta_prime is None:
            self.beta_prime = _xavier_uniform((features.shape[1], 1), rng)

    def fit(self, train_set, val_set=None):
        Recommender.fit(self, train_set, val_set)
        if getattr(train_set, "item_image", None) is None:
            raise CornacException("item_image modality is required but None.")
        features = np.asarray(train_set.item_image.features[: self.total_items]).astype(np.float32)
        self._init(self.total_users, self.total_items, features)
        trainer = _lib.VbprTrainer(features, self.total_users, self.total_items, self.k, self.k2, device=self.device)
        try:
            trainer.set_params(Bi=self.beta_item, Gu=self.gamma_user, Gi=self.gamma_item, Tu=self.theta_user,
                               E=self.emb_matrix, Bp=self.beta_prime)
            self.loss_history = []
            if self.trainable:
                for _ in range(self.n_epochs):
                    # the sampler is host-side in the reference too; one epoch of batches per device call
                    bu, bi, bj = [], [], []
                    for u, i, j in train_set.uij_iter(self.batch_size, shuffle=True):
                        bu.append(u); bi.append(i); bj.append(j)
                    u, i, j = np.concatenate(bu), np.concatenate(bi), np.concatenate(bj)
                    nll = trainer.fit_batches(u, i, j, self.batch_size, self.learning_rate, self.lambda_w,
                                              self.lambda_b, self.lambda_e)
                    self.loss_history.append(nll / len(u))
                if self.verbose:
                    print("Optimization finished!")
            p = trainer.get_params()
            self.beta_item, self.gamma_user, self.gamma_item = p["Bi"], p["Gu"], p["Gi"]
            self.theta_user, self.emb_matrix, self.beta_prime = p["Tu"], p["E"], p["Bp"].reshape(-1, 1)
            self.theta_item, self.visual_bias = trainer.item_tables()  # recom_vbpr.py:273-274
        finally:
            trainer.close()
        self._drop_scorer()
        return self

    # score(u, i) = beta_i + visual_bias_i + <gamma_u, gamma_i> + <theta_u, theta_item_i>  (recom_vbpr.py:277-304):
    # one dot product over the concatenated [gamma | theta] tables
    def _scoring_tables(self):
        if self.__dict__.get("_cat_src") is not self.gamma_user:
            self._cat_u = np.ascontiguousarray(np.concatenate([self.gamma_user, self.theta_user], axis=1), np.float32)
            self._cat_i = np.ascontiguousarray(np.concatenate([self.gamma_item, self.theta_item], axis=1), np.float32)
            self._cat_b = (self.beta_item + self.visual_bias).astype(np.float32)
            self._cat_src = self.gamma_user
        return self._cat_u, self._cat_i, self._cat_b, None

    def _drop_scorer(self):
        for name in ("_cat_u", "_cat_i", "_cat_b", "_cat_src"):  # derived tables follow the parameters they copy
            self.__dict__.pop(name, None)
        super()._drop_scorer()

    def _scorer_row_count(self):
        return len(self.gamma_user)

    def score(self, user_idx, item_idx=None):
        if item_idx is None:
            return self._get_scorer().score_user(user_idx)
        s = self.beta_item[item_idx] + self.visual_bias[item_idx]
        s += np.dot(self.gamma_item[item_idx], self.gamma_user[user_idx])
        s += np.dot(self.theta_item[item_idx], self.theta_user[user_idx])
        return s
