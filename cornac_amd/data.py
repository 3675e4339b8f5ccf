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

    def idx_iter(self, idx_range, batch_size=1, shuffle=False):
        """dataset.py:418-443: one `rng.shuffle` of arange(idx_range), then consecutive slices"""
        indices = np.arange(idx_range)
        if shuffle:
            self.rng.shuffle(indices)
        for b in range(int(np.ceil(len(indices) / batch_size))):
            yield indices[batch_size * b: min(batch_size * b + batch_size, len(indices))]

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


class ImageFeatures:
    """Minimal stand-in for cornac.data.ImageModality: item visual features [n_items, d]."""

    def __init__(self, features):
        self.features = np.asarray(features)


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
