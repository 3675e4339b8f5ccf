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
        self.U, self.V = (np.array(x, dtype=np.float32, order="C") for x in (U, V))
        self.B = None if B is None else np.array(B, dtype=np.float32, order="C")

    def set_factors_f64(self, U, V, B):
        self.U, self.V, self.B = (np.array(x, dtype=np.float64, order="C") for x in (U, V, B))

    def get_factors_f64(self):
        assert self.U.dtype == np.float64
        return self.U.copy(), self.V.copy(), self.B.copy()

    def fit_epochs_f64(self, n_epochs, lr, reg, use_bias=True, neg_population=_lib.NEG_UNIFORM):
        assert self.U.dtype == np.float64
        nnz = len(self.user_ids)
        popularity = neg_population == _lib.NEG_POPULARITY
        neg_ids = self.indices if popularity else np.arange(self.n_items, dtype=np.int32)
        neg_hi = nnz - 1 if popularity else self.n_items - 1
        correct = skipped = 0
        for _ in range(n_epochs):
            c, s = C.c_int64(), C.c_int64()
            rc = orc.lib().oracle_bpr_epoch_seq_f64(self.gp.ptr, self.gn.ptr, nnz - 1, neg_hi, nnz, self.user_ids,
                                                    self.indices, neg_ids, self.indptr, self.U, self.V, self.B, self.k,
                                                    lr, reg, int(use_bias), C.byref(c), C.byref(s))
            assert rc == 0
            correct, skipped = correct + c.value, skipped + s.value
        return correct, skipped

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

    # VEBPR (recom_vebpr.pyx): view CSR + a third stream
    def set_views(self, view_indptr, view_indices):
        self.v_indptr = np.ascontiguousarray(view_indptr, np.int32)
        self.v_indices = np.ascontiguousarray(view_indices, np.int32)
        if len(self.v_indices) == 0:
            self.v_indices = np.zeros(1, np.int32)

    def seed_view_stream(self, seed_view):
        self.calls.append(("view", int(seed_view)))
        self.gv = orc.MT19937(seed_view)

    def fit_epochs_vebpr(self, n_epochs, lr, reg, alpha, mode=_lib.MODE_HOGWILD):
        assert mode == _lib.MODE_DETERMINISTIC, "the double runs VEBPR's seeded path only"
        correct = skipped = 0
        for _ in range(n_epochs):
            c, s = C.c_int64(), C.c_int64()
            orc.lib().oracle_vebpr_epoch_seq(self.gp.ptr, self.gv.ptr, self.gn.ptr, len(self.user_ids), self.n_items,
                                             self.user_ids, self.indices, self.indptr, self.v_indices, self.v_indptr,
                                             self.U, self.V, self.k, lr, reg, alpha, C.byref(c), C.byref(s))
            correct, skipped = correct + c.value, skipped + s.value
        return correct, skipped

    def fit_epochs_vebpr_f64(self, n_epochs, lr, reg, alpha):
        assert self.U.dtype == np.float64
        correct = skipped = 0
        for _ in range(n_epochs):
            c, s = C.c_int64(), C.c_int64()
            orc.lib().oracle_vebpr_epoch_seq_f64(self.gp.ptr, self.gv.ptr, self.gn.ptr, len(self.user_ids), self.n_items,
                                                 self.user_ids, self.indices, self.indptr, self.v_indices, self.v_indptr,
                                                 self.U, self.V, self.k, lr, reg, alpha, C.byref(c), C.byref(s))
            correct, skipped = correct + c.value, skipped + s.value
        return correct, skipped

    def last_timing(self):
        return {}

    def get_factors(self):
        return self.U.copy(), self.V.copy(), None if self.B is None else self.B.copy()

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

    def reset_optimizer(self):
        self.steps = []
        self.keeps = None

    def fit_minibatch(self, order, batch_size, optimizer, lr, reg, mu, use_bias=True, keep_u=None, keep_i=None,
                      keep_scale=1.0):
        """the double re-runs the torch optimiser from the initial parameters over every batch seen so far (optimiser
        state has no other home here); returns the last epoch's sum of squared errors like the device call"""
        from oracle import mf_minibatch_oracle

        if not self.steps:
            self.start = (self.U.copy(), self.V.copy(), self.Bu.copy(), self.Bi.copy())
        order = np.asarray(order)
        batches = [order[b:b + batch_size] for b in range(0, len(order), batch_size)]
        self.steps += batches
        if keep_u is not None:
            self.keeps = (getattr(self, "keeps", None) or []) + [(keep_u[b:b + batch_size], keep_i[b:b + batch_size])
                                                       for b in range(0, len(order), batch_size)]
        U, V, Bu, Bi, losses = mf_minibatch_oracle.fit(*self.start, mu, self.rid, self.cid, self.val, self.steps, optimizer,
                                                       lr, reg, use_bias, keep=getattr(self, "keeps", None),
                                                       keep_scale=keep_scale)
        self.U, self.V, self.Bu, self.Bi = U, V, Bu, Bi
        return float(np.sum(losses[-len(batches):]))

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

    def set_f64(self, U, V, item_base=None, user_base=None):
        self.f64 = tuple(None if a is None else np.array(a, np.float64) for a in (U, V, item_base, user_base))

    def score_user_f64(self, user):
        U, V, ib, ub = self.f64
        out = V @ U[int(user)]
        return out + (0.0 if ib is None else ib) + (0.0 if ub is None else ub[int(user)])

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

    def set_exclusions(self, indptr=None, indices=None):
        self._excl = None if indptr is None else (np.asarray(indptr, np.int64), np.asarray(indices, np.int32))

    def rank_topk_resident(self, users, topk, fetch=True, timed=False, pinned=False):
        users = np.asarray(users, np.int32)
        ip, ix = self._excl
        cnt = ip[users + 1] - ip[users]
        ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        idx = np.concatenate([ix[ip[u]:ip[u + 1]] for u in users] + [np.zeros(0, np.int32)]).astype(np.int32)
        items, scores = self.rank_topk(users, topk, exclude=(ptr, idx))
        return (items, None if fetch == "items" else scores)

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


class FakeWmfTrainer:
    def __init__(self, csc, k, device=0):
        self.csc, self.k = csc, int(k)

    def set_factors(self, U, V):
        self.U0, self.V0 = np.array(U, np.float32), np.array(V, np.float32)
        self.oracle = None

    def fit_batches(self, batches, lambda_u, lambda_v, a, b, lr):
        from oracle.wmf_oracle import WmfOracle

        if self.oracle is None:   # Adam moments live in the oracle object across epochs, as they do on the device
            self.oracle = WmfOracle(self.U0, self.V0, self.csc, lambda_u, lambda_v, a, b, lr)
        return np.array(self.oracle.fit_batches(batches))

    def get_factors(self):
        return self.oracle.U.copy(), self.oracle.V.copy()

    def close(self):
        pass


class FakeVbprTrainer:
    """the reference's torch minibatch body (recom_vbpr.py:228-262, restated in oracle/vbpr_oracle.py) with the
    optimiser state kept across calls, one call per epoch of batches like the device entry point"""
    NAMES = _lib.VbprTrainer.NAMES

    def __init__(self, features, n_users, n_items, k, k2, device=0):
        import torch

        self.F = torch.tensor(np.asarray(features, np.float32))
        self.P, self.opt = None, None

    def set_params(self, **params):
        import torch

        self.P = {n: torch.tensor(np.asarray(params[n], np.float32).reshape(-1, 1) if n == "Bp" else
                                  np.asarray(params[n], np.float32), requires_grad=True) for n in self.NAMES}

    def fit_batches(self, u, i, j, batch_size, lr, lambda_w, lambda_b, lambda_e):
        import torch

        P = self.P
        if self.opt is None:
            self.opt = torch.optim.Adam([P[n] for n in self.NAMES], lr=lr)
        l2 = lambda *ts: sum(t.pow(2).sum() for t in ts) / 2   # noqa: E731
        total = 0.0
        for b in range(0, len(u), batch_size):
            bu, bi, bj = (torch.as_tensor(np.asarray(x[b:b + batch_size], np.int64)) for x in (u, i, j))
            gu, tu, gi, gj = P["Gu"][bu], P["Tu"][bu], P["Gi"][bi], P["Gi"][bj]
            beta_i, beta_j = P["Bi"][bi], P["Bi"][bj]
            fd = self.F[bi] - self.F[bj]
            X = beta_i - beta_j + (gu * (gi - gj)).sum(dim=1) + (tu * fd.mm(P["E"])).sum(dim=1) + fd.mm(P["Bp"])
            ll = torch.nn.functional.logsigmoid(X).sum()
            reg = (l2(gu, gi, gj, tu) * lambda_w + l2(beta_i) * lambda_b + l2(beta_j) * lambda_b / 10
                   + l2(P["E"], P["Bp"]) * lambda_e)
            loss = -ll + reg
            self.opt.zero_grad()
            loss.backward()
            self.opt.step()
            total += float(loss.data.item())
        return total

    def get_params(self):
        return {n: self.P[n].data.numpy().copy().reshape(-1) if n == "Bp" else self.P[n].data.numpy().copy()
                for n in self.NAMES}

    def item_tables(self):
        return self.F.mm(self.P["E"]).data.numpy(), self.F.mm(self.P["Bp"]).data.numpy().ravel()

    def close(self):
        pass


def install(monkeypatch):
    """route the model classes' device calls to the doubles for the duration of one test"""
    orc.build()
    monkeypatch.setattr(_lib, "BprTrainer", FakeBprTrainer)
    monkeypatch.setattr(_lib, "MfTrainer", FakeMfTrainer)
    monkeypatch.setattr(_lib, "WmfTrainer", FakeWmfTrainer)
    monkeypatch.setattr(_lib, "VbprTrainer", FakeVbprTrainer)
    monkeypatch.setattr(_lib, "Scorer", FakeScorer)
