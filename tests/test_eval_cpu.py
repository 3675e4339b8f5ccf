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
