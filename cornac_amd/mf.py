"""Biased matrix factorisation on MI355X — constructor, `fit/score/rank` surface and learned
attributes (`u_factors, i_factors, u_biases, i_biases, global_mean`) of the reference's
`cornac.models.MF` (cornac/models/mf/recom_mf.py:29-302).  Two backends next to the reference's
`"cpu"` / `"pytorch"` switch (recom_mf.py:177-183):

  * `backend="hip"`            replaces `backend_cpu.fit_sgd` (recom_mf.py:189-209) by `cornac_hip_mf_fit`;
  * `backend="hip-minibatch"`  replaces `backend_pt.learn` (recom_mf.py:211-252, backend_pt.py:67-106: batches of
                               `batch_size` ratings, `optimizer` in {sgd, adam, rmsprop, adagrad} with
                               weight_decay = lambda_reg over the dense tables) by `cornac_hip_mf_fit_minibatch`;
                               `dropout > 0` (backend_pt.py:42,59) by `cornac_hip_mf_fit_minibatch_dropout` with the keep
                               masks drawn on the host as the reference's CPU run draws them (`_DropoutMasks`).
"""
import numpy as np

from . import _lib
from .recommender import Recommender, ScoreException, _table_fingerprint

DTYPE = np.float32


def _normal(shape, rng, std):
    # cornac/utils/init_utils.py:60-82 `normal(shape, mean=0, std, dtype=float32)`
    return rng.normal(0.0, std, shape).astype(DTYPE)


class _DropoutMasks:
    """The keep masks of `nn.Dropout(p)` on a batch's gathered user rows and item rows (backend_pt.py:42,59), drawn as the
    reference's run on the CPU draws them: `_fit_pt` seeds torch's generator (recom_mf.py:221-222), the model's two — with
    biases four — nn.Embedding tables consume it with their normal_ initialisation before their weights are replaced
    (backend_pt.py:45-53), then every batch draws bernoulli_(1 - p) for the user rows and again for the item rows, kept
    entries scaled by 1 / (1 - p) in float32.  torch is the generator here, nothing else of it is used; the arithmetic of
    the step stays on the device (cornac_hip_mf_fit_minibatch_dropout)."""

    def __init__(self, p, seed, n_users, n_items, k, use_bias):
        import torch

        self.torch, self.p, self.k = torch, float(p), int(k)
        if seed is not None:
            torch.manual_seed(seed)
        for shape in [(n_users, k), (n_items, k)] + ([(n_users, 1), (n_items, 1)] if use_bias else []):
            torch.empty(shape).normal_()
        # p = 1: F.dropout multiplies by zero without drawing (nothing kept, no scale)
        self.scale = float(np.float32(1.0) / np.float32(1.0 - self.p)) if self.p < 1.0 else 0.0

    def epoch(self, n, batch_size):
        """keyword arguments of MfTrainer.fit_minibatch for an epoch of n ratings in consecutive batches"""
        keep_u, keep_i = np.zeros((n, self.k), np.uint8), np.zeros((n, self.k), np.uint8)
        if self.p < 1.0:
            for b0 in range(0, n, batch_size):
                b = min(batch_size, n - b0)
                keep_u[b0:b0 + b] = self.torch.empty(b, self.k).bernoulli_(1.0 - self.p).numpy()
                keep_i[b0:b0 + b] = self.torch.empty(b, self.k).bernoulli_(1.0 - self.p).numpy()
        return {"keep_u": keep_u, "keep_i": keep_i, "keep_scale": self.scale}


class MF(Recommender):
    """Parameters are those of the reference (recom_mf.py:32-131); `backend` accepts "hip" and
    "hip-minibatch" (the reference raises ValueError for an unknown backend, recom_mf.py:183 — so does this).
    `mode` as in BPR: None -> deterministic when seeded, hogwild otherwise.

    One behaviour differs from the reference on purpose: a hogwild fit whose loss becomes non-finite RAISES (`HipError`,
    "diverged") after the tables have been overwritten, where `fit_sgd` (backend_cpu.pyx:62-97) returns the NaN model
    silently; the sequential (seeded) mode reproduces the reference, NaNs included.  Very popular items no longer get there
    by themselves: rows above 0.1 % of the ratings of a large problem train through copies merged every phase / launch
    (csrc/mf_blocks.inc "virtual rows")."""

    def __init__(self, name="MF", k=10, backend="hip", optimizer="sgd", max_iter=20, learning_rate=0.01,
                 batch_size=256, lambda_reg=0.02, dropout=0.0, use_bias=True, early_stop=False, num_threads=0,
                 trainable=True, verbose=False, init_params=None, seed=None, mode=None, device=0):
        super().__init__(name=name, trainable=trainable, verbose=verbose)
        self.k = k
        self.backend = backend
        self.optimizer = optimizer
        self.max_iter = max_iter
        self.learning_rate = learning_rate
        self.batch_size = batch_size
        self.lambda_reg = lambda_reg
        self.dropout = dropout
        self.use_bias = use_bias
        self.early_stop = early_stop
        self.seed = seed
        self.num_threads = num_threads
        if mode not in (None, "deterministic", "hogwild"):
            raise ValueError(f"mode={mode} is not supported")
        self.mode = mode
        self.device = device
        self.init_params = {} if init_params is None else init_params
        self.u_factors = self.init_params.get("U", None)
        self.i_factors = self.init_params.get("V", None)
        self.u_biases = self.init_params.get("Bu", None)
        self.i_biases = self.init_params.get("Bi", None)

    @property
    def effective_mode(self):
        if self.mode is not None:
            return self.mode
        return "deterministic" if self.seed is not None else "hogwild"

    def _init(self):
        # recom_mf.py:138-156 — a FRESH RandomState(seed) per fit; sizes use num_users/num_items
        rng = np.random.RandomState(self.seed)
        if self.u_factors is None:
            self.u_factors = _normal([self.num_users, self.k], rng, 0.01)
        if self.i_factors is None:
            self.i_factors = _normal([self.num_items, self.k], rng, 0.01)
        self.u_biases = np.zeros(self.num_users, dtype=DTYPE) if self.u_biases is None else self.u_biases
        self.i_biases = np.zeros(self.num_items, dtype=DTYPE) if self.i_biases is None else self.i_biases
        self.global_mean = np.dtype(DTYPE).type(self.global_mean if self.use_bias else 0.0)

    def fit(self, train_set, val_set=None):
        Recommender.fit(self, train_set, val_set)
        self._init()
        if self.trainable:
            if self.backend == "hip":
                self._fit_hip(train_set, val_set)
            elif self.backend == "hip-minibatch":
                self._fit_minibatch(train_set, val_set)
            else:
                raise ValueError(f"{self.backend} is not supported")
        self._drop_scorer()
        return self

    def _fit_hip(self, train_set, val_set):
        rid, cid, val = train_set.uir_tuple
        mode = _lib.MODE_DETERMINISTIC if self.effective_mode == "deterministic" else _lib.MODE_HOGWILD
        trainer = _lib.MfTrainer(rid, cid, val.astype(DTYPE), self.num_users, self.num_items, self.k,
                                 device=self.device)
        try:
            trainer.set_factors(self.u_factors, self.i_factors, self.u_biases, self.i_biases)
            self.loss_history, self.epochs_run = trainer.fit(self.max_iter, self.learning_rate, self.lambda_reg,
                                                             float(self.global_mean), self.use_bias, self.early_stop,
                                                             mode)
            self.last_timing = trainer.last_timing()
            U, V, Bu, Bi = trainer.get_factors()
            self.u_factors[...] = U
            self.i_factors[...] = V
            self.u_biases[...] = Bu
            self.i_biases[...] = Bi
        finally:
            trainer.close()
        if self.verbose:
            print("Optimization finished!")

    def _fit_minibatch(self, train_set, val_set):
        if self.optimizer not in _lib.MfTrainer.OPTIMIZERS:
            raise KeyError(self.optimizer)  # OPTIMIZER_DICT[optimizer], backend_pt.py:79
        if self.dropout < 0.0 or self.dropout > 1.0:  # nn.Dropout.__init__ (backend_pt.py:42)
            raise ValueError("dropout probability has to be between 0 and 1, but got {}".format(self.dropout))
        masks = _DropoutMasks(self.dropout, self.seed, self.num_users, self.num_items, self.k, self.use_bias) \
            if self.dropout != 0.0 else None
        rid, cid, val = train_set.uir_tuple
        trainer = _lib.MfTrainer(rid, cid, val.astype(DTYPE), self.num_users, self.num_items, self.k,
                                 device=self.device)
        try:
            trainer.set_factors(self.u_factors, self.i_factors, self.u_biases, self.i_biases)
            trainer.reset_optimizer()
            self.loss_history = []
            for _ in range(self.max_iter):
                # backend_pt.py:85-88: uir_iter(batch_size, shuffle=True) = consecutive slices of one shuffle
                order = np.concatenate(list(train_set.idx_iter(len(val), self.batch_size, shuffle=True)))
                keep = masks.epoch(len(order), self.batch_size) if masks is not None else {}
                sse = trainer.fit_minibatch(order, self.batch_size, self.optimizer, self.learning_rate,
                                            self.lambda_reg, float(self.global_mean), self.use_bias, **keep)
                self.loss_history.append(sse / len(val))
            U, V, Bu, Bi = trainer.get_factors()
            self.u_factors, self.i_factors = U, V
            if self.use_bias:
                self.u_biases, self.i_biases = Bu, Bi
        finally:
            trainer.close()

    # ---- prediction -------------------------------------------------------------------------------
    def _scoring_tables(self):
        src = (_table_fingerprint(self.i_biases), float(self.global_mean))
        if self.__dict__.get("_item_base") is None or self._item_base_src != src:
            self._item_base = (self.global_mean + self.i_biases).astype(DTYPE)
            self._item_base_src = src
        return self.u_factors, self.i_factors, self._item_base, self.u_biases

    def _drop_scorer(self):
        # the biases are refreshed IN PLACE by a refit: the derived item_base table must go with the device scorer
        for name in ("_item_base", "_item_base_src"):
            self.__dict__.pop(name, None)
        super()._drop_scorer()

    def score(self, user_idx, item_idx=None):
        """recom_mf.py:254-286"""
        if item_idx is not None and self.is_unknown_item(item_idx):
            raise ScoreException("Can't make score prediction for item %d" % item_idx)
        if item_idx is None:
            if self.knows_user(user_idx):
                return self._get_scorer().score_user(user_idx)
            return self.global_mean + self.i_biases
        item_score = self.global_mean + self.i_biases[item_idx]
        if self.knows_user(user_idx):
            item_score += self.u_biases[user_idx]
            item_score += self.u_factors[user_idx].dot(self.i_factors[item_idx])
        return item_score

    def get_vector_measure(self):
        return "dot"

    def get_user_vectors(self):
        return np.concatenate((self.u_factors, np.ones([self.u_factors.shape[0], 1])), axis=1)

    def get_item_vectors(self):
        return np.concatenate((self.i_factors, self.i_biases.reshape((-1, 1))), axis=1)
