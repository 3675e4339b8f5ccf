"""BPR and WBPR on MI355X — same constructor, `fit/score/rank` surface and learned attributes
(`u_factors`, `i_factors`, `i_biases`) as the reference models
(cornac/models/bpr/recom_bpr.pyx:65-333, cornac/models/bpr/recom_wbpr.pyx:30-144); the per-epoch
`_fit_sgd` call is replaced by libcornac_hip (include/cornac_hip.h).

Mode selection follows the reference's threading rule (recom_bpr.pyx:132-137): a `seed` forces
the sequential, reproducible path -> `deterministic` mode (bit-faithful mt19937 sample streams,
order-preserving level schedule); no seed -> `hogwild` mode, the counterpart of the reference's
racy multi-thread path.  `mode=` overrides the rule explicitly.
"""
import numpy as np

from . import _lib
from .recommender import Recommender

DTYPE = np.float32


def _uniform(shape, rng):
    # cornac/utils/init_utils.py:33-57 `uniform(shape, low=0, high=1, dtype=float32)`
    return rng.uniform(0.0, 1.0, shape).astype(DTYPE)


def rngvector_mt_seed(seed):
    """RNGVector(1, rows, seed): the single engine is mt19937(get_rng(seed).randint(2**31))
    (recom_bpr.pyx:55-59)."""
    return int(np.random.RandomState(seed).randint(2 ** 31))


class BPR(Recommender):
    """Bayesian Personalized Ranking (Rendle et al., UAI 2009).

    Parameters are those of the reference (recom_bpr.pyx:68-143) plus:

    mode: None | "deterministic" | "hogwild"
        None: deterministic if `seed` is given, else hogwild (the reference's num_threads rule).
    device: int, HIP device ordinal.
    """

    _neg_population = _lib.NEG_UNIFORM
    _shared_stream = False

    def __init__(self, name="BPR", k=10, max_iter=100, learning_rate=0.001, lambda_reg=0.01, use_bias=True,
                 num_threads=0, trainable=True, verbose=False, init_params=None, seed=None, mode=None, device=0):
        super().__init__(name=name, trainable=trainable, verbose=verbose)
        self.k = int(k)
        self.max_iter = max_iter
        self.learning_rate = learning_rate
        self.lambda_reg = lambda_reg
        self.use_bias = use_bias
        self.seed = seed
        self.rng = np.random.RandomState(seed)
        self.num_threads = num_threads  # kept for clone()/API compatibility; the device decides its own width
        if mode not in (None, "deterministic", "hogwild"):
            raise ValueError(f"mode={mode} is not supported")
        self.mode = mode
        self.device = device
        self.init_params = {} if init_params is None else init_params
        self.u_factors = self.init_params.get("U", None)
        self.i_factors = self.init_params.get("V", None)
        self.i_biases = self.init_params.get("Bi", None)

    @property
    def effective_mode(self):
        if self.mode is not None:
            return self.mode
        return "deterministic" if self.seed is not None else "hogwild"

    def _init(self):
        # recom_bpr.pyx:145-152 — sizes use total_users/total_items
        n_users, n_items = self.total_users, self.total_items
        if self.u_factors is None:
            self.u_factors = (_uniform((n_users, self.k), self.rng) - 0.5) / self.k
        if self.i_factors is None:
            self.i_factors = (_uniform((n_items, self.k), self.rng) - 0.5) / self.k
        if self.i_biases is None or self.use_bias is False:
            self.i_biases = np.zeros(n_items, dtype=DTYPE)
        # `_fit_sgd` is a fused-type (`floating`) function (recom_bpr.pyx:211-214): three float32 tables train in float,
        # three float64 tables (all given through init_params) in double; a mix fails its buffer check (ValueError)
        kinds = {np.asarray(a).dtype for a in (self.u_factors, self.i_factors, self.i_biases)}
        if len(kinds) != 1 or next(iter(kinds)) not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("Buffer dtype mismatch: U, V and Bi must all be float32 or all be float64 "
                             "(recom_bpr.pyx:211-214)")

    @property
    def trains_float64(self):
        return self.u_factors is not None and np.asarray(self.u_factors).dtype == np.float64

    def _seed_trainer(self, trainer):
        if self.effective_mode == "deterministic" or self.trains_float64:
            # recom_bpr.pyx:190-191: two draws from self.rng, in this order, AFTER _init
            seed_pos = rngvector_mt_seed(self.rng.randint(2 ** 31))
            seed_neg = rngvector_mt_seed(self.rng.randint(2 ** 31))
            trainer.seed_mt19937(seed_pos, seed_neg, shared_stream=False)
        else:
            lo, hi = int(self.rng.randint(2 ** 31)), int(self.rng.randint(2 ** 31))
            trainer.seed_hogwild((hi << 32) | lo)

    def fit(self, train_set, val_set=None):
        Recommender.fit(self, train_set, val_set)
        self._init()
        if not self.trainable:
            return self
        X = train_set.matrix
        if not X.has_sorted_indices:
            X.sort_indices()
        trainer = _lib.BprTrainer(X.indptr, X.indices, train_set.num_users, train_set.num_items, self.total_users,
                                  self.total_items, self.k, device=self.device)
        try:
            f64 = self.trains_float64
            if f64:
                # float64 tables: the sequential engine in double, whatever the mode (there is no float64 throughput
                # kernel; the reference's unseeded float64 run is its racy OpenMP loop, to which the sequential order
                # is one admissible interleaving)
                trainer.set_factors_f64(self.u_factors, self.i_factors, self.i_biases)
            else:
                trainer.set_factors(self.u_factors, self.i_factors, self.i_biases)
            self._seed_trainer(trainer)
            mode = _lib.MODE_DETERMINISTIC if self.effective_mode == "deterministic" else _lib.MODE_HOGWILD
            nnz = X.nnz
            self.fit_stats = []

            def run(n):
                if f64:
                    return trainer.fit_epochs_f64(n, self.learning_rate, self.lambda_reg, self.use_bias, self._neg_population)
                return trainer.fit_epochs(n, self.learning_rate, self.lambda_reg, self.use_bias, self._neg_population, mode)

            if self.verbose:
                from tqdm.auto import trange

                with trange(self.max_iter) as progress:
                    for _ in progress:
                        correct, skipped = run(1)
                        self.fit_stats.append((correct, skipped))
                        progress.set_postfix({
                            "correct": "%.2f%%" % (100.0 * correct / (nnz - skipped + 1e-8)),
                            "skipped": "%.2f%%" % (100.0 * skipped / nnz),
                        })
                print("Optimization finished!")
            else:
                correct, skipped = run(self.max_iter)
                self.fit_stats.append((correct, skipped))
            self.last_timing = trainer.last_timing()
            U, V, B = trainer.get_factors_f64() if f64 else trainer.get_factors()
            # the reference mutates the arrays in place (also user-provided init_params)
            self.u_factors[...] = U
            self.i_factors[...] = V
            self.i_biases[...] = B
        finally:
            trainer.close()
        self._drop_scorer()
        return self

    # ---- prediction -------------------------------------------------------------------------------
    def _scoring_tables(self):
        return self.u_factors, self.i_factors, self.i_biases, None

    def score(self, user_idx, item_idx=None):
        """recom_bpr.pyx:272-297: scores over len(i_biases) items, or one scalar."""
        if item_idx is None:
            if self.trains_float64:
                return self._get_scorer().score_user_f64(user_idx)
            return self._get_scorer().score_user(user_idx)
        return self.i_biases[item_idx] + np.dot(self.u_factors[user_idx], self.i_factors[item_idx])

    def _scorer_row_count(self):
        # a float64 model is not served by the float32 batched kernels (rank_batch, rank_topk, score_pairs): with no
        # device rows the evaluators and rank() take the per-user flow over score() — float64, like the reference's
        return 0 if self.trains_float64 else len(self.u_factors)

    # ANN mixin surface (recom_bpr.pyx:299-333)
    def get_vector_measure(self):
        return "dot"

    def get_user_vectors(self):
        return np.concatenate((self.u_factors, np.ones([self.u_factors.shape[0], 1])), axis=1)

    def get_item_vectors(self):
        return np.concatenate((self.i_factors, self.i_biases.reshape((-1, 1))), axis=1)


class WBPR(BPR):
    """Weighted BPR: negatives sampled proportionally to item popularity
    (cornac/models/bpr/recom_wbpr.pyx:30-144).  Same kernel; the negative population is
    `X.indices` and — as in the reference — ONE generator supplies both draws of a sample."""

    _neg_population = _lib.NEG_POPULARITY
    _shared_stream = True

    def __init__(self, name="WBPR", k=10, max_iter=100, learning_rate=0.001, lambda_reg=0.01, use_bias=True,
                 num_threads=0, trainable=True, verbose=False, init_params=None, seed=None, mode=None, device=0):
        super().__init__(name=name, k=k, max_iter=max_iter, learning_rate=learning_rate, lambda_reg=lambda_reg,
                         use_bias=use_bias, num_threads=num_threads, trainable=trainable, verbose=verbose,
                         init_params=init_params, seed=seed, mode=mode, device=device)

    def _seed_trainer(self, trainer):
        if self.effective_mode == "deterministic" or self.trains_float64:
            s = rngvector_mt_seed(self.rng.randint(2 ** 31))  # recom_wbpr.pyx:131
            trainer.seed_mt19937(s, s, shared_stream=True)
        else:
            lo, hi = int(self.rng.randint(2 ** 31)), int(self.rng.randint(2 ** 31))
            trainer.seed_hogwild((hi << 32) | lo)


class VEBPR(Recommender):
    """View-Enhanced BPR (Ding et al., TKDE 2019) — constructor, attributes (`u_factor`, `i_factor`) and
    `fit/score/rank` surface of cornac/models/bpr/recom_vebpr.pyx:43-380; `train_set` must be a
    PurchaseViewDataset (purchases in `matrix`, views in `view_matrix`)."""

    def __init__(self, name="VEBPR", k=10, max_iter=100, learning_rate=0.01, lambda_reg=0.1, num_threads=0,
                 trainable=True, verbose=False, init_params=None, seed=None, alpha=0.5, mode=None, device=0):
        super().__init__(name=name, trainable=trainable, verbose=verbose)
        self.k = int(k)
        self.max_iter = max_iter
        self.learning_rate = learning_rate
        self.lambda_reg = lambda_reg
        self.alpha = float(alpha)
        self.seed = seed
        self.rng = np.random.RandomState(seed)
        self.num_threads = num_threads
        if mode not in (None, "deterministic", "hogwild"):
            raise ValueError(f"mode={mode} is not supported")
        self.mode = mode
        self.device = device
        self.init_params = {} if init_params is None else init_params
        self.u_factor = self.init_params.get("U", None)
        self.i_factor = self.init_params.get("V", None)

    @property
    def effective_mode(self):
        if self.mode is not None:
            return self.mode
        return "deterministic" if self.seed is not None else "hogwild"

    def fit(self, train_set, val_set=None):
        Recommender.fit(self, train_set, val_set)
        if not hasattr(train_set, "view_matrix"):
            raise ValueError("VEBPR requires a PurchaseViewDataset. Build one with "
                             "PurchaseViewDataset.build(purchase_data, view_data) or "
                             "PurchaseViewDataset.attach_view(dataset, view_data).")
        self.view_matrix = train_set.view_matrix
        n_users, n_items = self.total_users, self.total_items
        if self.u_factor is None:
            self.u_factor = (_uniform((n_users, self.k), self.rng) - 0.5) / self.k
        if self.i_factor is None:
            self.i_factor = (_uniform((n_items, self.k), self.rng) - 0.5) / self.k
        # _fit_sgd_viewloss is a fused-type function (recom_vebpr.pyx:219): float32 tables train in float, two float64
        # tables (both given through init_params) in double; a mix fails its buffer check (ValueError) before anything moves
        kinds = {np.asarray(self.u_factor).dtype, np.asarray(self.i_factor).dtype}
        if len(kinds) != 1 or next(iter(kinds)) not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("Buffer dtype mismatch: U and V must both be float32 or both be float64 (got %s)"
                             % sorted(str(x) for x in kinds))
        f64 = next(iter(kinds)) == np.dtype(np.float64)
        if not self.trainable:
            return self
        X, Vw = train_set.matrix, train_set.view_matrix
        if not X.has_sorted_indices:
            X.sort_indices()
        trainer = _lib.BprTrainer(X.indptr, X.indices, train_set.num_users, train_set.num_items, n_users, n_items,
                                  self.k, device=self.device)
        try:
            trainer.set_views(Vw.indptr, Vw.indices)
            if f64:
                # float64 tables: the sequential engine in double, whatever the mode (as for BPR: there is no float64
                # throughput kernel, and the reference's unseeded float64 run is its racy loop, to which the sequential
                # order is one admissible interleaving)
                trainer.set_factors_f64(self.u_factor, self.i_factor, np.zeros(n_items, np.float64))
            else:
                trainer.set_factors(self.u_factor, self.i_factor, None)
            if self.effective_mode == "deterministic" or f64:
                # recom_vebpr.pyx:191-193: rng_pos, rng_view, rng_neg drawn in this order
                sp = rngvector_mt_seed(self.rng.randint(2 ** 31))
                sv = rngvector_mt_seed(self.rng.randint(2 ** 31))
                sn = rngvector_mt_seed(self.rng.randint(2 ** 31))
                trainer.seed_mt19937(sp, sn, shared_stream=False)
                trainer.seed_view_stream(sv)
                mode = _lib.MODE_DETERMINISTIC
            else:
                lo, hi = int(self.rng.randint(2 ** 31)), int(self.rng.randint(2 ** 31))
                trainer.seed_hogwild((hi << 32) | lo)
                mode = _lib.MODE_HOGWILD
            if f64:
                self.fit_stats = [trainer.fit_epochs_vebpr_f64(self.max_iter, self.learning_rate, self.lambda_reg, self.alpha)]
                U, V, _ = trainer.get_factors_f64()
            else:
                self.fit_stats = [trainer.fit_epochs_vebpr(self.max_iter, self.learning_rate, self.lambda_reg,
                                                           self.alpha, mode)]
                U, V, _ = trainer.get_factors()
            self.u_factor[...] = U
            self.i_factor[...] = V
        finally:
            trainer.close()
        self._drop_scorer()
        return self

    def _scoring_tables(self):
        return self.u_factor, self.i_factor, None, None

    @property
    def trains_float64(self):
        return self.u_factor is not None and np.asarray(self.u_factor).dtype == np.float64

    def _scorer_row_count(self):
        # (a float64 model takes the per-user flow over score(), like BPR's: the batched kernels are float32)
        return 0 if self.trains_float64 else len(self.u_factor)

    def score(self, user_idx, item_idx=None):
        if item_idx is None:
            if self.trains_float64:
                # the reference allocates a float32 output here whatever the tables' type and its fast_dot then refuses the
                # float64 tables (recom_vebpr.pyx:356-357): a float64 VEBPR trains and serves score(user, item), not this
                raise ValueError("Buffer dtype mismatch, expected 'double' but got 'float'")
            return self._get_scorer().score_user(user_idx)
        return np.dot(self.u_factor[user_idx], self.i_factor[item_idx])
