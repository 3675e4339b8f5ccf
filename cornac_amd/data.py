"""Interaction dataset handed to the trainers — the host-side mirror of the reference's
`cornac.data.Dataset` (cornac/data/dataset.py:31-358), restricted to what the hot path reads:

  * `uir_tuple`  (int64 users, int64 items, float64 ratings, in insertion order — dataset.py:340-344),
  * `matrix` / `csr_matrix`  (scipy CSR, int32 indptr/indices sorted per row — dataset.py:226-235),
  * `num_users`, `num_items`, `uid_map`, `iid_map`, `min_rating`, `max_rating`, `global_mean`,
  * `reset()` (re-seeds the dataset RNG, dataset.py:401-404).

A real `cornac.data.Dataset` exposes the same attributes, so the models in this package accept
either (duck typing); nothing here imports the reference.
"""
import warnings
from collections import OrderedDict

import numpy as np
from scipy.sparse import csr_matrix


class Dataset:
    def __init__(self, num_users, num_items, uid_map, iid_map, uir_tuple, timestamps=None, seed=None):
        self.num_users = int(num_users)
        self.num_items = int(num_items)
        self.uid_map = uid_map
        self.iid_map = iid_map
        self.uir_tuple = uir_tuple
        self.timestamps = timestamps
        self.seed = seed
        self.rng = np.random.RandomState(seed)
        r = uir_tuple[2]
        self.num_ratings = len(r)
        self.max_rating = np.max(r)
        self.min_rating = np.min(r)
        self.global_mean = np.mean(r)
        self._csr = None

    @classmethod
    def build(cls, data, fmt="UIR", global_uid_map=None, global_iid_map=None, seed=None, exclude_unknowns=False):
        """(user, item, rating[, timestamp]) tuples -> Dataset; first occurrence of a (user, item) pair wins,
        ids are numbered in order of first appearance (dataset.py:257-358)."""
        fmt = fmt.upper()
        if fmt not in ("UIR", "UIRT"):
            raise ValueError("fmt should be in ['UIR', 'UIRT']")
        gu = OrderedDict() if global_uid_map is None else global_uid_map
        gi = OrderedDict() if global_iid_map is None else global_iid_map
        if len(data) >= cls.VECTORISED_BUILD_FROM:
            built = cls._build_columns(data, fmt, gu, gi, exclude_unknowns)
            if built is not None:
                uir, ts, dups = built
                if dups:
                    warnings.warn("%d duplicated observations are removed!" % dups)
                if len(uir[0]) == 0:
                    raise ValueError("data is empty after being filtered!")
                return cls(len(gu), len(gi), gu, gi, uir, timestamps=ts, seed=seed)
        seen = set()
        us, its, rs, ts = [], [], [], []
        dups = 0
        for rec in data:
            uid, iid, rating = rec[0], rec[1], rec[2]
            if exclude_unknowns and (uid not in gu or iid not in gi):
                continue
            if (uid, iid) in seen:
                dups += 1
                continue
            seen.add((uid, iid))
            us.append(gu.setdefault(uid, len(gu)))
            its.append(gi.setdefault(iid, len(gi)))
            rs.append(float(rating))
            if fmt == "UIRT":
                ts.append(int(rec[3]))
        if dups:
            warnings.warn("%d duplicated observations are removed!" % dups)
        if not seen:
            raise ValueError("data is empty after being filtered!")
        uir = (np.asarray(us, dtype="int"), np.asarray(its, dtype="int"), np.asarray(rs, dtype="float"))
        return cls(len(gu), len(gi), gu, gi, uir, timestamps=np.asarray(ts, dtype="int") if fmt == "UIRT" else None,
                   seed=seed)

    VECTORISED_BUILD_FROM = 20000   # records; below that the record-by-record loop above is as fast

    @staticmethod
    def _build_columns(data, fmt, gu, gi, exclude_unknowns):
        """the loop of `build` column-wise, for large inputs (the reference's loop takes about a minute on 20 M tuples):
        ids are factorised in order of first appearance — the order the loop assigns them in, since a skipped duplicate
        never introduces a new id — mapped through / appended to the global maps, and the first record of every
        (user, item) pair is kept.  Returns None when the ids cannot be factorised faithfully (then the loop runs)."""
        try:
            import pandas as pd
        except ImportError:
            return None
        from operator import itemgetter

        width = 4 if fmt == "UIRT" else 3
        if len(data[0]) < width:
            return None
        cols = [list(map(itemgetter(c), data)) for c in range(width)]
        idx = []
        for raw, gmap in ((cols[0], gu), (cols[1], gi)):
            values = np.empty(len(raw), dtype=object)
            values[:] = raw
            codes, uniques = pd.factorize(values)
            if (codes < 0).any():        # NaN-like ids: pandas treats them as missing, a dict would not
                return None
            mapped = np.empty(len(uniques), dtype=np.int64)
            for n, key in enumerate(uniques):
                known = gmap.get(key)
                if known is None:
                    known = -1 if exclude_unknowns else gmap.setdefault(key, len(gmap))
                mapped[n] = known
            idx.append(mapped[codes])
        u_idx, i_idx = idx
        kept = np.flatnonzero((u_idx >= 0) & (i_idx >= 0))
        pair = u_idx[kept] * np.int64(max(len(gi), 1)) + i_idx[kept]
        _, first = np.unique(pair, return_index=True)
        first = np.sort(first)
        rows = kept[first]
        ratings = np.asarray(cols[2], dtype="float")[rows]
        ts = np.asarray(cols[3], dtype="int")[rows] if fmt == "UIRT" else None
        uir = (u_idx[rows].astype("int"), i_idx[rows].astype("int"), ratings)
        return uir, ts, len(kept) - len(first)

    @classmethod
    def from_uirt(cls, data, seed=None):
        return cls.build(data, fmt="UIRT", seed=seed)

    @classmethod
    def from_uir(cls, data, seed=None):
        return cls.build(data, seed=seed)

    @classmethod
    def from_arrays(cls, users, items, ratings, num_users=None, num_items=None, seed=None):
        """Index arrays -> Dataset without the Python-level id mapping loop (identity id maps);
        used for large synthetic interaction sets.  (user, item) pairs must be unique."""
        users = np.asarray(users, dtype="int")
        items = np.asarray(items, dtype="int")
        ratings = np.asarray(ratings, dtype="float")
        nu = int(users.max()) + 1 if num_users is None else int(num_users)
        ni = int(items.max()) + 1 if num_items is None else int(num_items)
        return cls(nu, ni, _RangeMap(nu), _RangeMap(ni), (users, items, ratings), seed=seed)

    def reset(self):
        self.rng = np.random.RandomState(self.seed)
        return self

    @property
    def matrix(self):
        return self.csr_matrix

    @property
    def csr_matrix(self):
        if self._csr is None:
            u, i, r = self.uir_tuple
            self._csr = csr_matrix((r, (u, i)), shape=(self.num_users, self.num_items))
        return self._csr

    @property
    def csc_matrix(self):
        """item-major view of the same ratings (dataset.py:247-255)"""
        if getattr(self, "_csc", None) is None:
            self._csc = self.csr_matrix.tocsc()
        return self._csc

    @property
    def user_ids(self):
        return list(self.uid_map.keys())

    @property
    def item_ids(self):
        return list(self.iid_map.keys())

    # ---- batch iterators used by minibatch models (VBPR) -------------------------------------------
    item_image = None  # optional modality: any object with a `.features` [n_items, d] array

    def num_batches(self, batch_size):
        return int(np.ceil(len(self.uir_tuple[0]) / batch_size))

    @property
    def dok(self):
        """{(user, item): rating} — what the reference's `dok_matrix` lookups return (dataset.py:237-245)"""
        if getattr(self, "_dok", None) is None:
            u, i, r = self.uir_tuple
            self._dok = dict(zip(zip(u.tolist(), i.tolist()), r.tolist()))
        return self._dok

    @property
    def dok_matrix(self):
        """the ratings as a scipy DOK matrix (dataset.py:247-255); the iterators use the plain dict `dok`"""
        if getattr(self, "_dok_matrix", None) is None:
            self._dok_matrix = self.csr_matrix.todok()
        return self._dok_matrix

    def _grouped(self, by, other, with_time):
        """{key: (others, ratings[, timestamps])} in data order per key (dataset.py:136-220); with_time: each key's
        lists sorted by np.argsort of its timestamps, exactly as the reference sorts them"""
        if with_time and self.timestamps is None:
            raise ValueError("Timestamps are required but None!")
        keys, vals, ratings = self.uir_tuple[by], self.uir_tuple[other], self.uir_tuple[2]
        order = np.argsort(keys, kind="stable")
        cuts = np.flatnonzero(np.diff(keys[order])) + 1
        out = {}
        first_seen = {}
        for sel in (np.split(order, cuts) if len(order) else []):
            cols = [vals[sel].tolist(), ratings[sel].tolist()]
            if with_time:
                times = self.timestamps[sel].tolist()
                chrono = np.argsort(times)
                cols = [[c[p] for p in chrono] for c in cols + [times]]
            first_seen[int(keys[sel[0]])] = (int(sel[0]), tuple(cols))
        for key, (_, cols) in sorted(first_seen.items(), key=lambda kv: kv[1][0]):   # keys in order of first appearance
            out[key] = cols
        return out

    @property
    def user_data(self):
        return self._grouped(0, 1, False)

    @property
    def item_data(self):
        return self._grouped(1, 0, False)

    @property
    def chrono_user_data(self):
        return self._grouped(0, 1, True)

    @property
    def chrono_item_data(self):
        return self._grouped(1, 0, True)

    def num_user_batches(self, batch_size):
        return int(np.ceil(self.num_users / batch_size))

    def num_item_batches(self, batch_size):
        return int(np.ceil(self.num_items / batch_size))

    def add_modalities(self, **kwargs):
        for name in ("user_feature", "item_feature", "user_text", "item_text", "user_image", "item_image", "user_graph",
                     "item_graph", "sentiment", "review_text"):
            setattr(self, name, kwargs.get(name, None))

    def save(self, fpath):
        """pickle the dataset (dataset.py:585-597)"""
        import copy
        import os
        import pickle

        os.makedirs(os.path.dirname(fpath) or ".", exist_ok=True)
        with open(fpath, "wb") as f:
            pickle.dump(copy.deepcopy(self), f, protocol=pickle.HIGHEST_PROTOCOL)

    @staticmethod
    def load(fpath):
        import pickle

        with open(fpath, "rb") as f:
            dataset = pickle.load(f)
        dataset.load_from = fpath
        return dataset

    def idx_iter(self, idx_range, batch_size=1, shuffle=False):
        """dataset.py:418-443: one `rng.shuffle` of arange(idx_range), then consecutive slices"""
        indices = np.arange(idx_range)
        if shuffle:
            self.rng.shuffle(indices)
        for b in range(int(np.ceil(len(indices) / batch_size))):
            yield indices[batch_size * b: min(batch_size * b + batch_size, len(indices))]

    def uir_iter(self, batch_size=1, shuffle=False, binary=False, num_zeros=0):
        """(users, items, ratings) batches (dataset.py:445-488); `num_zeros` unobserved items per observation are
        appended with rating 0, drawn like the reference draws them: `rng.randint(0, num_items)` until the pair has
        no positive rating"""
        users, items, ratings = self.uir_tuple
        dok = self.dok if num_zeros > 0 else None
        for batch_ids in self.idx_iter(len(users), batch_size, shuffle):
            bu, bi = users[batch_ids], items[batch_ids]
            br = np.ones_like(bi) if binary else ratings[batch_ids]
            if num_zeros > 0:
                rep = bu.repeat(num_zeros)
                neg = np.empty_like(rep)
                for t, u in enumerate(rep.tolist()):
                    j = self.rng.randint(0, self.num_items)
                    while dok.get((u, int(j)), 0.0) > 0:
                        j = self.rng.randint(0, self.num_items)
                    neg[t] = j
                bu, bi, br = np.concatenate((bu, rep)), np.concatenate((bi, neg)), np.concatenate((br, np.zeros_like(neg)))
            yield bu, bi, br

    def user_iter(self, batch_size=1, shuffle=False):
        """batches of user indices (dataset.py:528-544), candidate order as in `item_iter`"""
        user_indices = np.fromiter(set(self.uir_tuple[0].tolist()), dtype="int")
        for batch_ids in self.idx_iter(len(user_indices), batch_size, shuffle):
            yield user_indices[batch_ids]

    def item_iter(self, batch_size=1, shuffle=False):
        """batches of item indices (dataset.py:546-562); the candidate order is the iteration order of
        the set of observed items, as in the reference"""
        item_indices = np.fromiter(set(self.uir_tuple[1].tolist()), "int")
        for batch_ids in self.idx_iter(len(item_indices), batch_size, shuffle):
            yield item_indices[batch_ids]

    def uij_iter(self, batch_size=1, shuffle=False, neg_sampling="uniform"):
        """(users, positive items, negative items) batches, reproducing the reference's sampler
        draw for draw (dataset.py:490-526): per observation `rng.choice(neg_population)`, redrawn
        while the user's rating of the drawn item is >= the positive's rating."""
        if neg_sampling.lower() == "uniform":
            neg_population = np.arange(self.num_items)
        elif neg_sampling.lower() == "popularity":
            neg_population = self.uir_tuple[1]
        else:
            raise ValueError("Unsupported negative sampling option: {}".format(neg_sampling))
        dok = self.dok
        for batch_ids in self.idx_iter(len(self.uir_tuple[0]), batch_size, shuffle):
            batch_users = self.uir_tuple[0][batch_ids]
            batch_pos_items = self.uir_tuple[1][batch_ids]
            batch_pos_ratings = self.uir_tuple[2][batch_ids]
            batch_neg_items = np.empty_like(batch_pos_items)
            for t, (user, pos_rating) in enumerate(zip(batch_users.tolist(), batch_pos_ratings.tolist())):
                neg_item = self.rng.choice(neg_population)
                while dok.get((user, int(neg_item)), 0.0) >= pos_rating:
                    neg_item = self.rng.choice(neg_population)
                batch_neg_items[t] = neg_item
            yield batch_users, batch_pos_items, batch_neg_items


class FeatureModality:
    """Per-user / per-item feature rows (cornac/data/modality.py:41-113): `features` [n, d] aligned with the raw ids
    `ids`; `build(id_map)` moves each known id's row to its mapped index (rows of unknown ids stay where they were)
    and, if `normalized`, min-max scales the whole matrix — the order the reference does it in."""

    def __init__(self, features=None, ids=None, normalized=False, **kwargs):
        self.features = features
        self.ids = ids
        self.normalized = normalized

    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, value):
        if value is not None:
            assert len(value.shape) == 2
        self._features = value

    @property
    def feature_dim(self):
        return self.features.shape[1]

    def build(self, id_map=None, **kwargs):
        if self.features is None:
            return
        if self.ids is not None and id_map is not None:
            placed = [(id_map.get(raw), old) for old, raw in enumerate(self.ids)]
            new_rows = np.array([n for n, _ in placed if n is not None], dtype=np.int64)
            old_rows = np.array([o for n, o in placed if n is not None], dtype=np.int64)
            assert len(new_rows) == 0 or new_rows.max() < self.features.shape[0]
            moved, ids = np.copy(self.features), list(self.ids)
            moved[new_rows] = self.features[old_rows]
            for n, o in zip(new_rows.tolist(), old_rows.tolist()):
                ids[n] = self.ids[o]
            self.features, self.ids = moved, ids
        if self.normalized:
            self.features = self.features - np.min(self.features)
            self.features = self.features / (np.max(self.features) + 1e-10)
        return self

    def batch_feature(self, batch_ids):
        assert self.features is not None
        return self.features[batch_ids]


class ImageModality(FeatureModality):
    """item (or user) visual features, optionally with the raw images / their paths (cornac/data/image.py:20-83);
    VBPR reads `train_set.item_image.features`"""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.images = kwargs.get("images", None)
        self.paths = kwargs.get("paths", None)

    def batch_image(self, batch_ids, target_size=(256, 256), color_mode="rgb", interpolation="nearest"):
        raise NotImplementedError


class ImageFeatures(ImageModality):
    """features already in item-index order: `ImageFeatures(F)` == `ImageModality(features=F)`"""

    def __init__(self, features):
        super().__init__(features=np.asarray(features))


class _RangeMap:
    """Identity raw-id -> index map over range(n) that behaves like the OrderedDict id maps
    (len, get, keys, items, in) without materialising n Python objects."""

    def __init__(self, n):
        self.n = int(n)

    def __len__(self):
        return self.n

    def __contains__(self, key):
        return isinstance(key, (int, np.integer)) and 0 <= key < self.n

    def get(self, key, default=None):
        return int(key) if key in self else default

    def __getitem__(self, key):
        if key not in self:
            raise KeyError(key)
        return int(key)

    def keys(self):
        return range(self.n)

    def items(self):
        return ((i, i) for i in range(self.n))


class PurchaseViewDataset(Dataset):
    """Purchases + a secondary "view" matrix in one id space — mirror of
    cornac/data/dataset.py:1400-1521 (used by VEBPR).  View entries that are also purchases are
    dropped, the view CSR has sorted indices."""

    def __init__(self, dataset, view_matrix):
        super().__init__(dataset.num_users, dataset.num_items, dataset.uid_map, dataset.iid_map, dataset.uir_tuple,
                         timestamps=getattr(dataset, "timestamps", None), seed=getattr(dataset, "seed", None))
        view_matrix = view_matrix - view_matrix.multiply(self.matrix > 0)
        view_matrix.eliminate_zeros()
        view_matrix.sort_indices()
        self.view_matrix = view_matrix.tocsr()

    @classmethod
    def build(cls, purchase_data, view_data, seed=None):
        gu, gi = OrderedDict(), OrderedDict()
        purchase_set = Dataset.build(purchase_data, global_uid_map=gu, global_iid_map=gi, seed=seed)
        view_set = Dataset.build(view_data, global_uid_map=gu, global_iid_map=gi, seed=seed)
        full = Dataset(len(gu), len(gi), gu, gi, purchase_set.uir_tuple, seed=seed)
        return cls(full, view_set.matrix)

    @classmethod
    def attach_view(cls, dataset, view_data):
        view_set = Dataset.build(view_data, global_uid_map=dataset.uid_map, global_iid_map=dataset.iid_map,
                                 exclude_unknowns=True)
        return cls(dataset, view_set.matrix)
