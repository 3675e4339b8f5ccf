"""Model base class of the MI355X backend — host-side mirror of the reference's plug-in API
`cornac.models.Recommender` (cornac/models/recommender.py:84-653): `fit / score / rate / rank /
recommend / save / load / clone`, same names, argument meaning and error behaviour, so a model of
this package is used exactly like a reference model (and its parity tests read the same).

What differs by design: `rank()` and `rank_batch()` run on the device (batched users x items
scoring + top-k kernels in libcornac_hip) instead of `score()` + NumPy argsort per user
(recommender.py:503-530).  Learned parameters stay plain NumPy attributes (`u_factors`,
`i_factors`, ...), so pickling, `clone()` and ANN wrappers keep working; device handles live in
attributes listed in `ignored_attrs` (mechanism at recommender.py:137, :182-190).
"""
import copy
import inspect
import json
import os
import pickle
import warnings
from datetime import datetime
from glob import glob

import numpy as np


class _ExceptionRoot(Exception):
    """(a Python-level root so that adopt_reference_classes() can re-base the two exceptions below)"""


class CornacException(_ExceptionRoot):
    """cornac/exception.py:16-20"""


class ScoreException(CornacException):
    """Raised by score() for unknown users/items (cornac/exception.py:22-26)."""


class _RecommenderRoot:
    """(a Python-level root so that adopt_reference_classes() can re-base Recommender)"""


_adopted = False


def adopt_reference_classes():
    """Class identity with the reference, where the reference is importable (the boundary is the class).

    The reference keeps only `isinstance(model, cornac.models.Recommender)` objects in an `Experiment`
    (cornac/experiment/experiment.py:90-100), `BaseSearch` is itself a `Recommender` wrapping one
    (cornac/hyperopt.py:96-183), and its evaluators catch `cornac.exception.ScoreException`
    (cornac/eval_methods/base_method.py, cornac/models/recommender.py:447-474).  When `cornac.models.recommender` is
    loaded in this process, this re-bases `cornac_amd.Recommender` on the reference's class and the two exceptions on
    the reference's, so every cornac_amd model IS a `cornac.models.Recommender` and what it raises IS a
    `cornac.exception.ScoreException`.  Every method of the mirror overrides the reference's, so behaviour does not
    change.  Called at import and whenever a model is constructed; a no-op without the reference (the GPU box)."""
    global _adopted
    if _adopted:
        return True
    import sys

    rec = sys.modules.get("cornac.models.recommender")
    exc = sys.modules.get("cornac.exception")
    if rec is None or exc is None or not hasattr(rec, "Recommender"):
        return False
    try:
        CornacException.__bases__ = (exc.CornacException,)
        ScoreException.__bases__ = (exc.ScoreException, CornacException)
        Recommender.__bases__ = (rec.Recommender,)
    except (TypeError, AttributeError):  # an incompatible class layout: stay the standalone mirror
        return False
    _adopted = True
    return True


_PROBE = 64


def _table_fingerprint(a):
    """Identity AND a content probe of a host table: (id, shape, dtype, hash of <= 64 evenly spaced
    elements).  The device copy of the scoring tables is keyed on it, so re-assigned attributes and whole-table in-place
    edits (`m.i_biases += 1`, a warm-start loader writing into the existing arrays) are noticed without
    `invalidate_scorer()`; only a sparse edit that misses every probed element still needs that call.  A few
    microseconds per table (a strided view, no index arrays): it runs on every score() call."""
    if a is None:
        return None
    if type(a) is not np.ndarray:
        a = np.asarray(a)
    flat = a.reshape(-1) if a.flags.c_contiguous else a.ravel()
    n = flat.size
    probe = flat[:: max(1, n // _PROBE)][:_PROBE].tobytes() if n else b""
    return (id(a), a.shape, a.dtype.num, hash(probe))


def clip(values, lower_bound, upper_bound):
    """cornac/utils/common.py `clip`: clamp into [lower_bound, upper_bound]."""
    values = np.where(values > upper_bound, upper_bound, values)
    values = np.where(values < lower_bound, lower_bound, values)
    return values


class Recommender(_RecommenderRoot):
    # what prediction needs to remember about the training set (recommender.py:330-337)
    _DATASET_FACTS = ("num_users", "num_items", "uid_map", "iid_map", "min_rating", "max_rating", "global_mean")

    def __init__(self, name, trainable=True, verbose=False):
        adopt_reference_classes()
        self.name = name
        self.trainable = trainable
        self.verbose = verbose
        self.is_fitted = False
        # not pickled / deep-copied: datasets and device-side state
        self.ignored_attrs = ["train_set", "val_set", "test_set", "_scorer", "_scorer_key", "_trainer", "_item_base",
                              "_item_base_src", "_item_base_mean", "_cat_u", "_cat_i", "_cat_b", "_cat_src", "_excl_reg"]
        for attr in self._DATASET_FACTS:
            setattr(self, attr, None)
        self._item_ids = None

    # ---- bookkeeping -------------------------------------------------------------------------
    @property
    def total_users(self):
        return len(self.uid_map) if self.uid_map is not None else self.num_users

    @property
    def total_items(self):
        return len(self.iid_map) if self.iid_map is not None else self.num_items

    @property
    def user_ids(self):
        return list(self.uid_map.keys())

    @property
    def item_ids(self):
        if self._item_ids is None:
            self._item_ids = list(self.iid_map.keys())
        return self._item_ids

    def reset_info(self):
        self.best_value = float("-inf")
        self.best_epoch = 0
        self.current_epoch = 0
        self.stopped_epoch = 0
        self.wait = 0

    def __deepcopy__(self, memo):
        cls = self.__class__
        result = cls.__new__(cls)
        ignored = set(self.ignored_attrs)
        for k, v in self.__dict__.items():
            if k in ignored:
                continue
            setattr(result, k, copy.deepcopy(v))
        return result

    def __getstate__(self):
        ignored = set(self.ignored_attrs)
        return {k: v for k, v in self.__dict__.items() if k not in ignored}

    @classmethod
    def _get_init_params(cls):
        sig = inspect.signature(cls.__init__)
        return sorted(p.name for p in sig.parameters.values() if p.name != "self")

    def clone(self, new_params=None):
        """Fresh, unfitted instance with the same constructor arguments (recommender.py:204-221)."""
        new_params = {} if new_params is None else new_params
        init_params = {}
        for name in self._get_init_params():
            init_params[name] = new_params.get(name, copy.deepcopy(getattr(self, name)))
        return self.__class__(**init_params)

    def save(self, save_dir=None, save_trainset=False, metadata=None):
        """Pickle the model (+ .meta json, optional .trainset) under save_dir/<name>/ (recommender.py:223-276)."""
        if save_dir is None:
            return None
        folder = os.path.join(save_dir, self.name)
        os.makedirs(folder, exist_ok=True)
        model_file = os.path.join(folder, datetime.now().strftime("%Y-%m-%d_%H-%M-%S-%f") + ".pkl")

        def dump(obj, path):
            with open(path, "wb") as fh:
                pickle.dump(obj, fh, protocol=pickle.HIGHEST_PROTOCOL)

        dump(copy.deepcopy(self), model_file)  # deep copy drops ignored_attrs (datasets, device handles)
        meta = dict(metadata or {}, model_classname=type(self).__name__, model_file=os.path.basename(model_file))
        if save_trainset:
            dump(self.train_set, model_file + ".trainset")
            meta["trainset_file"] = os.path.basename(model_file) + ".trainset"
        with open(model_file + ".meta", "w", encoding="utf-8") as fh:
            json.dump(meta, fh, ensure_ascii=False, indent=4)
        if self.verbose:
            print("{} model is saved to {}".format(self.name, model_file))
        return model_file

    @staticmethod
    def load(model_path, trainable=False):
        """recommender.py:278-304 (a directory loads its newest .pkl)."""
        if os.path.isdir(model_path):
            model_file = sorted(glob("{}/*.pkl".format(model_path)))[-1]
        else:
            model_file = model_path
        with open(model_file, "rb") as f:
            model = pickle.load(f)
        model.trainable = trainable
        model.load_from = model_file
        return model

    def fit(self, train_set, val_set=None):
        """Record the dataset facts prediction needs (recommender.py:306-346)."""
        if self.is_fitted:
            warnings.warn("Model is already fitted. Re-fitting will overwrite the previous model.")
        self.reset_info()
        for ds in (train_set, val_set):
            if ds is not None:
                ds.reset()  # re-seed the dataset samplers for reproducibility
        for attr in self._DATASET_FACTS:
            setattr(self, attr, getattr(train_set, attr))
        self.train_set, self.val_set = train_set, val_set
        self.is_fitted = True
        self._item_ids = None
        self._drop_scorer()
        return self

    def transform(self, test_set):
        """hook called by the evaluation method before scoring a test set (recommender.py:410-421); nothing to cache here"""

    def knows_user(self, user_idx):
        return user_idx is not None and 0 <= user_idx < self.num_users

    def knows_item(self, item_idx):
        return item_idx is not None and 0 <= item_idx < self.num_items

    def is_unknown_user(self, user_idx):
        return not self.knows_user(user_idx)

    def is_unknown_item(self, item_idx):
        return not self.knows_item(item_idx)

    # ---- prediction --------------------------------------------------------------------------
    def score(self, user_idx, item_idx=None):
        raise NotImplementedError("The algorithm is not able to make score prediction!")

    def default_score(self):
        return self.global_mean

    def rate(self, user_idx, item_idx, clipping=True):
        """recommender.py:447-474"""
        try:
            rating_pred = self.score(user_idx, item_idx)
        except ScoreException:
            rating_pred = self.default_score()
        if clipping:
            rating_pred = clip(rating_pred, self.min_rating, self.max_rating)
        return rating_pred

    def rate_batch(self, user_indices, item_indices, clipping=True):
        """Batched rate(): predictions for many (user, item) pairs in one kernel — what
        `rating_eval` (cornac/eval_methods/base_method.py:35-105) computes one Python call at a
        time.  Pairs with an unknown user or item go through `rate()` itself."""
        u = np.asarray(user_indices, dtype=np.int64)
        i = np.asarray(item_indices, dtype=np.int64)
        sc = self._get_scorer()
        rows = self._scorer_rows(u)
        known = (rows >= 0) & (i >= 0) & (i < sc.n_items) & (i < self.num_items)
        out = np.empty(len(u), dtype=np.float64)
        if known.any():
            clip_rng = (self.min_rating, self.max_rating) if clipping else None
            out[known] = sc.score_pairs(rows[known], i[known], clip=clip_rng)
        for p in np.flatnonzero(~known):   # pairs outside the device tables keep the model's own rule (e.g. MF scores an
            out[p] = self.rate(int(u[p]), int(i[p]), clipping)   # unknown user with the item's bias, recom_mf.py:281-286)
        return out

    # device scorer -------------------------------------------------------------------------------
    def _scoring_tables(self):
        """(U, V, item_base, user_base) such that score(u, i) = item_base[i] + user_base[u] + <U[u], V[i]>."""
        raise NotImplementedError

    def _drop_scorer(self):
        sc = self.__dict__.pop("_scorer", None)
        self.__dict__.pop("_scorer_key", None)
        if sc is not None:
            sc.close()

    def invalidate_scorer(self):
        """Drop the device copy of the scoring tables.  The copy is keyed on identity plus a content probe of the host
        arrays (`_table_fingerprint`), so whole-table in-place edits are noticed on their own; call this after a SPARSE
        in-place edit (a handful of rows).  `fit()` calls it itself."""
        self._drop_scorer()

    def _get_scorer(self):
        from . import _lib

        U, V, ib, ub = self._scoring_tables()
        key = tuple(_table_fingerprint(x) for x in (U, V, ib, ub))
        if self.__dict__.get("_scorer") is None or self.__dict__.get("_scorer_key") != key:
            self._drop_scorer()
            self._scorer = _lib.Scorer(U, V, ib, ub, device=getattr(self, "device", 0))
            if np.asarray(U).dtype == np.float64:   # a model trained in double also serves score() in double
                self._scorer.set_f64(U, V, ib, ub)
            self._scorer_key = key
        return self._scorer

    def rank(self, user_idx, item_indices=None, k=-1, **kwargs):
        """Rank items for one user: `(ranked_items, item_scores)` with the reference's meaning
        (recommender.py:476-530): `item_scores[t]` is the score of `item_indices[t]`,
        `ranked_items` are the candidates by descending score.

        Scores (score_user kernel) and the ordering (top-k / sort kernels) run on the device.
        Ties are ordered by descending item index (the reference leaves tie order unspecified,
        tests/cornac/models/test_recommender.py:89-93).  With k != -1 the result holds, like the reference's
        (recommender.py:521-528), EVERY candidate: the first k are the ranked top-k, the remaining ones follow in
        candidate order (the reference leaves them in argpartition order) — callers such as the reference's
        `ranking_eval` hand the whole array to metrics that read past k.  A k larger than the number of candidates
        ranks all of them (the reference fails inside np.argpartition there)."""
        try:
            known_item_scores = self.score(user_idx, **kwargs)
        except ScoreException:
            known_item_scores = np.ones(self.total_items) * self.default_score()
        n_known = len(known_item_scores)
        if n_known == self.total_items:
            all_item_scores = known_item_scores
        else:
            all_item_scores = np.ones(self.total_items) * np.min(known_item_scores)
            all_item_scores[: self.num_items] = known_item_scores
        item_indices = np.arange(self.num_items) if item_indices is None else np.asarray(item_indices)
        item_scores = all_item_scores[item_indices]
        n_cand = len(item_indices)
        topk = n_cand if k == -1 else min(int(k), n_cand)
        row = self._scorer_row(user_idx)
        mask = None
        if row is not None and n_cand > 0 and int(item_indices.max()) < self._get_scorer().n_items:
            mask = np.ones(self._get_scorer().n_items, dtype=bool)
            mask[item_indices] = False
            if n_cand != mask.size - int(mask.sum()):
                mask = None   # repeated candidates: the reference returns them as given (recommender.py:515-530)
        if mask is not None:
            sc = self._get_scorer()
            excl = np.flatnonzero(mask).astype(np.int32)
            items, _ = sc.rank_topk(np.array([row], np.int32), topk,
                                    exclude=(np.array([0, len(excl)], np.int64), excl) if len(excl) else None)
            ranked_items = items[0].astype(item_indices.dtype)
            if topk < n_cand:  # the candidates outside the top-k follow, unranked (recommender.py:521-528)
                in_top = np.zeros(sc.n_items, dtype=bool)
                in_top[ranked_items] = True
                ranked_items = np.concatenate([ranked_items, item_indices[~in_top[item_indices]]])
        else:
            # user unknown to the device tables (constant scores), candidates beyond the scored items (all tied at the
            # row minimum) or repeated candidates: the scores are on the host already — order them there under the
            # pinned rule (descending score, ties by descending item index); every candidate is returned, repeats too
            order = np.lexsort((item_indices, item_scores))[::-1]
            ranked_items = item_indices[order]   # all of them ranked: more than k != -1 asks for
        return ranked_items, item_scores

    @property
    def batch_num_items(self):
        """number of items `rank_batch` / `rank_positions_batch` rank over: the rows of the device item table"""
        return len(self._scoring_tables()[1])

    def rank_batch(self, user_indices, k=10, exclude=None):
        """Batched top-k for many users in one scoring-GEMM + top-k pass (what the evaluation loop
        cornac/eval_methods/base_method.py:176-220 does one user at a time).

        exclude: optional CSR `(indptr int64[n+1], indices int32)` of items to drop per listed user
        (e.g. training positives).  Returns `(items [n, k] int32, scores [n, k] float32)`, padded
        with (-1, -inf) when a user has fewer than k candidates."""
        rows = self._scorer_rows(user_indices)
        if (rows < 0).any():
            raise ScoreException("rank_batch needs users known to the model")
        sc = self._get_scorer()
        topk = sc.n_items if k == -1 else min(int(k), sc.n_items)
        return sc.rank_topk(rows.astype(np.int32), topk, exclude=exclude)

    def register_exclusions(self, token, user_indices, ex_ptr, ex_idx):
        """Keep the per-user exclusion lists of an evaluation split on the device, once per (device scorer, split):
        `ex_ptr / ex_idx` is a CSR over `user_indices` (ascending users with a device row).  `token` identifies the
        split; a repeated call with the same token on the same scorer is free.  Returns False when the scorer cannot
        hold resident lists (the caller then passes the lists per call)."""
        sc = self._get_scorer()
        if not hasattr(sc, "set_exclusions"):
            return False
        self._excl_reg = (token, user_indices, ex_ptr, ex_idx)  # (a scorer rebuilt later re-registers from here)
        self._register_exclusions_on(sc)
        return True

    def _register_exclusions_on(self, sc):
        token, user_indices, ex_ptr, ex_idx = self._excl_reg
        if getattr(sc, "_excl_token", None) == token:
            return
        rows = self._scorer_rows(user_indices)
        if (rows < 0).any():
            raise ScoreException("register_exclusions needs users known to the model")
        counts = np.zeros(sc.n_users, dtype=np.int64)
        counts[rows] = np.diff(np.asarray(ex_ptr, dtype=np.int64))
        full_ptr = np.concatenate(([0], np.cumsum(counts)))
        # the listed users ascend, so their lists concatenated in that order ARE the full CSR's index array
        sc.set_exclusions(full_ptr, np.ascontiguousarray(ex_idx, dtype=np.int32))
        sc._excl_token = token

    def rank_batch_resident(self, user_indices, k=10):
        """`rank_batch` against the lists of `register_exclusions`: nothing but the user ids goes to the device and
        only the ranked item ids come back (page-locked buffer owned by the scorer: valid until the next call)."""
        rows = self._scorer_rows(user_indices)
        if (rows < 0).any():
            raise ScoreException("rank_batch needs users known to the model")
        if self.__dict__.get("_excl_reg") is None:
            raise ScoreException("rank_batch_resident needs register_exclusions first")
        sc = self._get_scorer()
        self._register_exclusions_on(sc)  # (free unless the scorer was rebuilt since)
        topk = min(int(k), sc.n_items)
        items, _ = sc.rank_topk_resident(rows.astype(np.int32), topk, fetch="items", pinned=True)
        return items

    def rank_positions_batch(self, user_indices, targets, exclude=None):
        """Where each listed item stands in its user's ranking, without producing the rankings: `targets` and
        `exclude` are CSR `(indptr int64[n+1], indices int32)` per listed user (test positives / training
        positives in `ranking_eval`).  Returns per target `(greater, pos, ge, score)` — candidates scored strictly
        higher, 0-based position in the order `rank()` gives, candidates scored at least as high, its score."""
        rows = self._scorer_rows(user_indices)
        if (rows < 0).any():
            raise ScoreException("rank_positions_batch needs users known to the model")
        return self._get_scorer().rank_positions(rows.astype(np.int32), targets, exclude=exclude)

    def _scorer_row_count(self):
        """users 0 .. count-1 are rows of the device user table (models trained over `total_users` override this)"""
        return self.num_users

    def _scorer_row(self, user_idx):
        """Row of the device user table for user_idx, or None if its score is not table-driven."""
        return int(user_idx) if user_idx is not None and 0 <= user_idx < self._scorer_row_count() else None

    def _scorer_rows(self, user_indices):
        """`_scorer_row` for an array of users: int64 rows, -1 where there is none"""
        u = np.asarray(user_indices, dtype=np.int64)
        return np.where((u >= 0) & (u < self._scorer_row_count()), u, -1)

    def recommend(self, user_id, k=-1, remove_seen=False, train_set=None):
        """Top-k raw item ids for a raw user id (recommender.py:532-580)."""
        user_idx = self.uid_map.get(user_id, -1)
        if user_idx == -1:
            raise ValueError(f"{user_id} is unknown to the model.")
        if not -1 <= k <= self.total_items:
            raise ValueError(f"k={k} is invalid, there are {self.total_users} users in total.")
        candidates = np.arange(self.total_items)
        if remove_seen:
            if train_set is None:
                raise ValueError("train_set must be provided to remove seen items.")
            X = train_set.csr_matrix
            if user_idx < X.shape[0]:
                candidates = np.setdiff1d(candidates, X.indices[X.indptr[user_idx]:X.indptr[user_idx + 1]])
        ranked, _ = self.rank(user_idx, candidates, k=k)   # device top-k (the reference sorts everything, then cuts)
        if k != -1:
            ranked = ranked[:k]
        names = self.item_ids
        return [names[i] for i in ranked]


adopt_reference_classes()
