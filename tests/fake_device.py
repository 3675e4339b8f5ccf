"""TEST DOUBLE of the device layer — stand-ins for `cornac_amd._lib.BprTrainer / MfTrainer / Scorer` that run the CPU
oracle, so the `-m "not gpu"` suite can drive the HOST logic of the model classes (seed derivation order, factor
initialisation and copy-back, mode selection, the Recommender surface, the evaluation and experiment layers) without a
GPU.  Test infrastructure only: nothing under cornac_amd/ imports this, and it proves nothing about the HIP kernels
(the `-m gpu` tests do that through the real C ABI)."""
import ctypes as C

import numpy as np

from cornac_amd import _lib
from oracle import oracle as orc


class FakeBprTrainer:
    def __init__(self, indptr, indices, n_users, n_items, total_users, total_items, k, device=0):
        self.indptr = np.ascontiguousarray(indptr, np.int32)
        self.indices = np.ascontiguousarray(indices, np.int32)
        self.user_ids = np.repeat(np.arange(n_users), np.diff(self.indptr)).astype(np.int32)
        self.n_items, self.k = int(n_items), int(k)
        self.gp = self.gn = None
        self.hog_seed = None
        self.calls = []

    def set_factors(self, U, V, B):
        self.U, self.V, self.B = (np.array(x, dtype=np.float32, order="C") for x in (U, V, B))

    def seed_mt19937(self, seed_pos, seed_neg, shared_stream=False):
        self.calls.append(("mt19937", int(seed_pos), int(seed_neg), bool(shared_stream)))
        self.gp = orc.MT19937(seed_pos)
        self.gn = self.gp if shared_stream else orc.MT19937(seed_neg)

    def seed_hogwild(self, seed):
        self.calls.append(("hogwild", int(seed)))
        self.hog_seed = int(seed)

    def fit_epochs(self, n_epochs, lr, reg, use_bias, neg_population, mode, flags=0):
        nnz = len(self.user_ids)
        correct = skipped = 0
        if mode == _lib.MODE_DETERMINISTIC:
            popularity = neg_population == _lib.NEG_POPULARITY
            neg_ids = self.indices if popularity else np.arange(self.n_items, dtype=np.int32)
            neg_hi = nnz - 1 if popularity else self.n_items - 1
            for _ in range(n_epochs):
                c, s = C.c_int64(), C.c_int64()
                rc = orc.lib().oracle_bpr_epoch_seq(self.gp.ptr, self.gn.ptr, nnz - 1, neg_hi, nnz, self.user_ids,
                                                    self.indices, neg_ids, self.indptr, self.U, self.V, self.B, self.k,
                                                    lr, reg, int(use_bias), C.byref(c), C.byref(s), None, None, None)
                assert rc == 0
                correct, skipped = correct + c.value, skipped + s.value
        else:
            assert neg_population == _lib.NEG_UNIFORM, "the double's hogwild path draws uniform negatives only"
            correct, skipped = orc.bpr_hogwild_epochs(self.indptr, self.indices, self.user_ids, self.n_items, self.U, self.V,
                                                      self.B, self.k, lr, reg, use_bias, self.hog_seed % (2 ** 31), 2,
                                                      n_epochs, fast=False)
        return correct, skipped

    def last_timing(self):
        return {}

    def get_factors(self):
        return self.U.copy(), self.V.copy(), self.B.copy()

    def close(self):
        pass


class FakeMfTrainer:
    OPTIMIZERS = _lib.MfTrainer.OPTIMIZERS

    def __init__(self, rid, cid, val, n_users, n_items, k, device=0):
        self.rid = np.ascontiguousarray(rid, np.int64)
        self.cid = np.ascontiguousarray(cid, np.int64)
        self.val = np.ascontiguousarray(val, np.float32)
        self.k = int(k)

    def set_factors(self, U=None, V=None, Bu=None, Bi=None):
        self.U, self.V, self.Bu, self.Bi = (np.array(x, dtype=np.float32, order="C") for x in (U, V, Bu, Bi))

    def fit(self, max_iter, lr, reg, mu, use_bias=True, early_stop=False, mode=_lib.MODE_HOGWILD):
        loss = np.zeros(max(max_iter, 1), np.float32)
        threads = 1 if mode == _lib.MODE_DETERMINISTIC else 2
        epochs = orc.lib().oracle_mf_fit(self.rid, self.cid, self.val, len(self.val), self.U, self.V, self.Bu, self.Bi, self.k,
                                         lr, reg, float(mu), max_iter, threads, int(use_bias), int(early_stop),
                                         loss.ctypes.data)
        return loss[:epochs], epochs

    def last_timing(self):
        return {}

    def get_factors(self):
        return self.U.copy(), self.V.copy(), self.Bu.copy(), self.Bi.copy()

    def close(self):
        pass


class FakeScorer:
    """scores = the oracle's fma-chain dot product (what the device kernels reproduce bit for bit); order = descending
    score, ties by higher item index (the pinned tie rule)"""

    def __init__(self, U, V, item_base=None, user_base=None, device=0):
        self.U, self.V = np.ascontiguousarray(U, np.float32), np.ascontiguousarray(V, np.float32)
        self.ib = None if item_base is None else np.ascontiguousarray(item_base, np.float32)
        self.ub = None if user_base is None else np.ascontiguousarray(user_base, np.float32)
        self.n_users, self.k = self.U.shape
        self.n_items = self.V.shape[0]

    def close(self):
        pass

    def score_block(self, users):
        return orc.score_block(self.U, self.V, self.ib, self.ub, np.asarray(users, np.int32))

    def score_user(self, user):
        return self.score_block([int(user)])[0]

    def score_pairs(self, users, items, clip=None):
        users, items = np.asarray(users), np.asarray(items)
        out = np.empty(len(users), np.float32)
        for u in np.unique(users):
            sel = users == u
            out[sel] = self.score_user(u)[items[sel]]
        return out if clip is None else np.clip(out, np.float32(clip[0]), np.float32(clip[1]))

    def _candidates(self, exclude, r):
        if exclude is None:
            return np.arange(self.n_items)
        return np.setdiff1d(np.arange(self.n_items), exclude[1][exclude[0][r]:exclude[0][r + 1]])

    def rank_topk(self, users, topk, exclude=None):
        items = np.full((len(users), topk), -1, np.int32)
        scores = np.full((len(users), topk), -np.inf, np.float32)
        S = self.score_block(users)
        for r in range(len(users)):
            cand = self._candidates(exclude, r)
            order = cand[np.argsort(S[r][cand], kind="stable")[::-1]][:topk]
            items[r, :len(order)], scores[r, :len(order)] = order, S[r][order]
        return items, scores

    def rank_positions(self, users, targets, exclude=None):
        S = self.score_block(users)
        out = [[], [], [], []]
        for r in range(len(users)):
            cand = self._candidates(exclude, r)
            sc = S[r][cand]
            for t in targets[1][targets[0][r]:targets[0][r + 1]]:
                s = S[r][t]
                out[0].append((sc > s).sum())
                out[1].append((sc > s).sum() + ((sc == s) & (cand > t)).sum())
                out[2].append((sc >= s).sum())
                out[3].append(s)
        return tuple(np.array(o, d) for o, d in zip(out, (np.int32, np.int32, np.int32, np.float32)))


def install(monkeypatch):
    """route the model classes' device calls to the doubles for the duration of one test"""
    orc.build()
    monkeypatch.setattr(_lib, "BprTrainer", FakeBprTrainer)
    monkeypatch.setattr(_lib, "MfTrainer", FakeMfTrainer)
    monkeypatch.setattr(_lib, "Scorer", FakeScorer)
