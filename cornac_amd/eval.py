"""Batched evaluation loops over the device scoring kernels.

`ranking_eval` and `rating_eval` keep the contract of the reference's
cornac/eval_methods/base_method.py:108-226 / :35-105 — same arguments, same masks, same metric
`compute(...)` calls, same return value `(avg_results, user_results)` — but replace the
one-`rank()`-per-user / one-`rate()`-per-rating Python loops by
`rank_batch` (fused scoring GEMM + top-k with the training/validation positives as per-user
exclusion lists) and `rate_batch` (one gather-dot-clip kernel).

Metrics whose `k` is -1 (AUC, MAP, MRR need the full ranked list / all scores) fall back to the
reference's per-user flow through `model.rank(user, item_indices, k=-1)`, which is still
device-scored and device-sorted.
"""
import numpy as np


def _pos_items(csr, user_idx, threshold):
    if user_idx >= csr.shape[0]:
        return np.empty(0, dtype=np.int64)
    lo, hi = csr.indptr[user_idx], csr.indptr[user_idx + 1]
    return csr.indices[lo:hi][csr.data[lo:hi] >= threshold].astype(np.int64)


def ranking_eval(model, metrics, train_set, test_set, val_set=None, rating_threshold=1.0, exclude_unknowns=True,
                 verbose=False, batch_users=16384):
    if len(metrics) == 0:
        return [], []
    max_k = max(m.k for m in metrics)
    need_full = any(m.k <= 0 for m in metrics)
    test_mat, train_mat = test_set.csr_matrix, train_set.csr_matrix
    val_mat = None if val_set is None else val_set.csr_matrix
    n_eval_items = train_set.num_items if exclude_unknowns else test_set.num_items
    user_results = [{} for _ in metrics]

    users, gt_pos, excl = [], [], []
    for user_idx in sorted(set(int(u) for u in test_set.uir_tuple[0])):
        tp = _pos_items(test_mat, user_idx, rating_threshold)
        if len(tp) == 0:
            continue
        vp = np.empty(0, np.int64) if val_mat is None else _pos_items(val_mat, user_idx, rating_threshold)
        trp = _pos_items(train_mat, user_idx, rating_threshold)
        tp_eval = tp[tp < n_eval_items]
        # candidates = test positives + everything that is in no positive list; i.e. exclude the
        # train/val positives that are not also test positives (base_method.py:188-206)
        ex = np.setdiff1d(np.union1d(vp, trp), tp)
        users.append(user_idx)
        gt_pos.append(np.sort(tp_eval))
        excl.append(ex[ex < n_eval_items].astype(np.int32))

    if need_full or not hasattr(model, "rank_batch"):
        all_items = np.arange(n_eval_items)
        for user_idx, gp, ex in zip(users, gt_pos, excl):
            item_indices = np.setdiff1d(all_items, ex)
            gt_neg = np.setdiff1d(item_indices, gp)
            rank_, scores_ = model.rank(user_idx=user_idx, item_indices=item_indices, k=max_k if not need_full else -1)
            for i, mt in enumerate(metrics):
                user_results[i][user_idx] = mt.compute(gt_pos=gp, gt_neg=gt_neg, pd_rank=rank_, pd_scores=scores_,
                                                       item_indices=item_indices)
    else:
        for b0 in range(0, len(users), batch_users):
            ub = users[b0:b0 + batch_users]
            eb = excl[b0:b0 + batch_users]
            indptr = np.concatenate([[0], np.cumsum([len(e) for e in eb])]).astype(np.int64)
            indices = np.concatenate(eb).astype(np.int32) if indptr[-1] else np.empty(0, np.int32)
            items, _ = model.rank_batch(ub, k=max_k, exclude=(indptr, indices))
            for r, user_idx in enumerate(ub):
                pd_rank = items[r][items[r] >= 0].astype(np.int64)
                for i, mt in enumerate(metrics):
                    user_results[i][user_idx] = mt.compute(gt_pos=gt_pos[b0 + r], gt_neg=None, pd_rank=pd_rank,
                                                           pd_scores=None, item_indices=None)
    avg_results = [sum(ur.values()) / len(ur) for ur in user_results]
    return avg_results, user_results


def rating_eval(model, metrics, test_set, user_based=False, verbose=False):
    """base_method.py:35-105 with one batched prediction kernel for all test ratings."""
    if len(metrics) == 0:
        return [], []
    u_indices, i_indices, r_values = test_set.uir_tuple
    r_preds = model.rate_batch(u_indices, i_indices)
    avg_results, user_results = [], []

    def scalar(v):
        return v.item() if hasattr(v, "item") else v

    for mt in metrics:
        if user_based:
            per_user = {}
            for user_idx in np.unique(u_indices):
                sel = u_indices == user_idx
                per_user[int(user_idx)] = scalar(mt.compute(gt_ratings=r_values[sel], pd_ratings=r_preds[sel]))
            user_results.append(per_user)
            avg_results.append(sum(per_user.values()) / len(per_user))
        else:
            user_results.append({})
            avg_results.append(scalar(mt.compute(gt_ratings=r_values, pd_ratings=r_preds)))
    return avg_results, user_results
