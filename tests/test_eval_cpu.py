"""CPU test of the vectorised evaluation-list building (cornac_amd.eval.eval_lists) against the per-user
numpy-set formulation of the reference's ranking_eval (base_method.py:176-206)."""
import numpy as np
import pytest
import scipy.sparse as sp

from cornac_amd.eval import eval_lists, eval_lists_loop


@pytest.mark.parametrize("seed,with_val,thr", [(0, False, 1.0), (1, True, 3.0), (2, True, 1.0)])
def test_eval_lists_equal_the_per_user_formulation(seed, with_val, thr):
    rs = np.random.RandomState(seed)
    nu, ni = 120, 80

    def rand_csr(n_rows, n_cols, nnz):
        keys = np.unique(rs.randint(0, n_rows * n_cols, nnz))
        return sp.csr_matrix((rs.randint(1, 6, len(keys)).astype(float), (keys // n_cols, keys % n_cols)), shape=(n_rows, n_cols))

    train = rand_csr(nu, ni, 1500)
    test = rand_csr(nu - 7, ni + 5, 600)        # fewer users, items unknown to the training set
    val = rand_csr(nu - 3, ni, 400) if with_val else None
    n_eval = ni
    users, gp, gi, ep, ei = eval_lists(train, test, val, thr, n_eval)
    test_users = np.repeat(np.arange(test.shape[0]), np.diff(test.indptr))
    users_l, gt_l, ex_l = eval_lists_loop(train, test, val, thr, n_eval, test_users)
    assert users.tolist() == users_l and len(users) > 50
    for r in range(len(users)):
        assert np.array_equal(gi[gp[r]:gp[r + 1]], gt_l[r])
        assert np.array_equal(ei[ep[r]:ep[r + 1]], ex_l[r])


def test_batched_metrics_equal_the_per_user_metrics():
    from cornac_amd import metrics as mm

    rs = np.random.RandomState(4)
    n_users, n_items, K = 300, 60, 20
    hits_rows, n_gt, gts, ranks = [], [], [], []
    for u in range(n_users):
        gt = np.sort(rs.choice(n_items, rs.randint(0, 12), replace=False))
        n_pred = rs.randint(0, K + 1)                         # users with fewer candidates than K (padded with -1)
        rank = rs.permutation(n_items)[:n_pred]
        gts.append(gt); ranks.append(rank); n_gt.append(len(gt))
        row = np.zeros(K, bool)
        row[:n_pred] = np.isin(rank, gt)
        hits_rows.append(row)
    hits, n_gt = np.array(hits_rows), np.array(n_gt)
    for cls in (mm.Recall, mm.Precision, mm.NDCG, mm.HitRatio):
        for k in (1, 5, 10, 20):
            mt = cls(k=k)
            got = mt.compute_batch(hits, n_gt)
            with np.errstate(divide="ignore", invalid="ignore"):
                want = np.array([mt.compute(gt_pos=gts[u], pd_rank=ranks[u]) for u in range(n_users)], dtype=float)
            assert np.allclose(got, want, rtol=1e-12, atol=0, equal_nan=True), (cls.__name__, k)


def test_fmeasure_and_ncrr_batch_forms_equal_per_user():
    from cornac_amd import metrics as mm

    rs = np.random.RandomState(5)
    n_users, n_items, K = 300, 60, 20
    gts, ranks, hits = [], [], np.zeros((n_users, K), bool)
    for u in range(n_users):
        gts.append(np.sort(rs.choice(n_items, rs.randint(1, 12), replace=False)))
        ranks.append(rs.permutation(n_items)[:rs.randint(1, K + 1)])
        hits[u, :len(ranks[u])] = np.isin(ranks[u], gts[u])
    n_gt, n_pred = np.array([len(g) for g in gts]), np.array([len(r) for r in ranks])
    for k in (1, 5, 10, 20):
        got = mm.NCRR(k=k).compute_batch(hits, n_gt, n_pred=n_pred)
        want = [mm.NCRR(k=k).compute(gt_pos=gts[u], pd_rank=ranks[u]) for u in range(n_users)]
        assert np.allclose(got, want, rtol=1e-12, atol=0), k
        full = n_pred >= k                                      # F1 (like precision) divides by k, so lists must reach k
        got = mm.FMeasure(k=k).compute_batch(hits, n_gt)
        want = [mm.FMeasure(k=k).compute(gt_pos=gts[u], pd_rank=ranks[u]) for u in range(n_users)]
        assert np.allclose(got[full], np.array(want, dtype=float)[full], rtol=1e-12, atol=0), k


@pytest.mark.parametrize("levels", [0, 3, 1])
def test_full_list_metric_batch_forms_equal_per_user_with_ties(levels):
    """AUC / MAP / MRR over ragged full candidate lists; `levels` > 0 quantises the scores so most of them tie"""
    from cornac_amd import metrics as mm

    rs = np.random.RandomState(6 + levels)
    n_users, n_items = 120, 70
    L = n_items
    hits, scores = np.zeros((n_users, L), bool), np.full((n_users, L), -np.inf, np.float32)
    n_cand, n_gt, per_user = np.zeros(n_users, np.int64), np.zeros(n_users, np.int64), []
    for u in range(n_users):
        cand = np.sort(rs.choice(n_items, rs.randint(5, n_items + 1), replace=False))
        gt = rs.choice(cand, rs.randint(1, 4), replace=False)
        sc = rs.normal(size=len(cand)).astype(np.float32)
        if levels:
            sc = np.round(sc * levels) / np.float32(levels)
        order = np.argsort(sc, kind="stable")[::-1]
        hits[u, :len(cand)] = np.isin(cand[order], gt)
        scores[u, :len(cand)] = sc[order]
        n_cand[u], n_gt[u] = len(cand), len(gt)
        per_user.append(dict(item_indices=cand, pd_scores=sc, gt_pos=gt, gt_neg=np.setdiff1d(cand, gt),
                             pd_rank=cand[order]))
    for cls in (mm.AUC, mm.MAP, mm.MRR):
        mt = cls()
        got = mt.compute_full_batch(hits, scores, n_cand, n_gt)
        want = np.array([mt.compute(**kw) for kw in per_user], dtype=float)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-15), cls.__name__
    for cls in (mm.NDCG, mm.Recall, mm.Precision, mm.NCRR, mm.FMeasure, mm.HitRatio):   # their k = -1 forms
        for rank_len in (None, 7):
            mt = cls(k=-1)
            got = mt.compute_full_batch(hits, scores, n_cand, n_gt, rank_len=rank_len)
            want = np.array([mt.compute(gt_pos=kw["gt_pos"], pd_rank=kw["pd_rank"][:rank_len]) for kw in per_user],
                            dtype=float)
            assert np.allclose(got, want, rtol=1e-12, atol=1e-15), (cls.__name__, rank_len)


def test_eval_lists_shortcuts_and_memo():
    """implicit feedback (every stored rating passes the threshold), square shapes where nothing is cut, and the
    per-split memo: the same matrix OBJECTS give the cached lists, equal-valued new objects are recomputed, and the
    result never aliases the caller's matrices"""
    from cornac_amd import eval as ev

    rs = np.random.RandomState(5)
    nu, ni = 90, 60

    def ones_csr(nnz):
        keys = np.unique(rs.randint(0, nu * ni, nnz))
        return sp.csr_matrix((np.ones(len(keys)), (keys // ni, keys % ni)), shape=(nu, ni))

    train, test = ones_csr(1200), ones_csr(2500)     # dense enough for every user to have a test positive
    test_users = np.repeat(np.arange(nu), np.diff(test.indptr))
    first = eval_lists(train, test, None, 1.0, ni)
    users_l, gt_l, ex_l = eval_lists_loop(train, test, None, 1.0, ni, test_users)
    users, gp, gi, ep, ei = first
    assert users.tolist() == users_l == list(range(nu))
    for r in range(nu):
        assert np.array_equal(gi[gp[r]:gp[r + 1]], gt_l[r]) and np.array_equal(ei[ep[r]:ep[r + 1]], ex_l[r])
    assert eval_lists(train, test, None, 1.0, ni) is first                 # memo hit: same objects, same arguments
    assert eval_lists(train, test, None, 2.0, ni) is not first             # another threshold
    again = eval_lists(train.copy(), test, None, 1.0, ni)                   # another object with the same values
    assert again is not first and all(np.array_equal(a, b) for a, b in zip(again, first))
    test2 = test.copy()
    test2.data[:] = 1.0
    before = test2.indices.copy()
    eval_lists(train, test2, None, 1.0, ni)
    assert np.array_equal(test2.indices, before) and len(ev._LISTS_CACHE) <= 4
