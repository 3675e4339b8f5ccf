"""Batched evaluation loops over the device scoring kernels.

`ranking_eval` and `rating_eval` keep the contract of the reference's
cornac/eval_methods/base_method.py:108-226 / :35-105 — same arguments, same masks, same metric
`compute(...)` calls, same return value `(avg_results, user_results)` — but replace the
one-`rank()`-per-user / one-`rate()`-per-rating Python loops by
`rank_batch` (fused scoring GEMM + top-k with the training/validation positives as per-user
exclusion lists) and `rate_batch` (one gather-dot-clip kernel).

Metrics over the full candidate list (AUC, MAP, MRR) are batched as well.  They only depend on where each test
positive stands among the user's candidates and on the length of its run of tied scores, so a model that offers
`rank_positions_batch` (cornac_hip_rank_positions: counts over the score tile, no sort, no ranking copied back) is
asked for exactly that; other models with `rank_batch` hand over full rankings of a block of users (score tile +
per-row sort) and the same positional, tie-aware vectorised forms (`compute_full_batch`) are applied to them.
Metric objects without batched forms (e.g. the reference's own classes) go through the reference's per-user flow,
`model.rank(user, item_indices, k)`, which is still device-scored and device-sorted.
"""
import numpy as np
from .metrics import positive_runs


def _pos_items(csr, user_idx, threshold):
    if user_idx >= csr.shape[0]:
        return np.empty(0, dtype=np.int64)
    lo, hi = csr.indptr[user_idx], csr.indptr[user_idx + 1]
    return csr.indices[lo:hi][csr.data[lo:hi] >= threshold].astype(np.int64)


def _bool_csr(mat, threshold, shape):
    """entries >= threshold of a CSR matrix as a boolean CSR of the given (larger or equal) shape — filtered in
    place of the stored order (row pointers from a running count of the kept entries), no COO round trip"""
    from scipy.sparse import csr_matrix

    if not mat.has_sorted_indices:
        mat = mat.sorted_indices()
    keep = mat.data >= threshold
    indptr = np.empty(shape[0] + 1, dtype=np.int64)
    if keep.all():   # implicit feedback / every rating at or above the threshold: the stored structure as it is
        indptr[: mat.shape[0] + 1] = mat.indptr
        indices, n_kept = mat.indices, mat.nnz
    else:
        kept_before = np.concatenate(([0], np.cumsum(keep, dtype=np.int64)))
        indptr[: mat.shape[0] + 1] = kept_before[mat.indptr]
        indices, n_kept = mat.indices[keep], int(kept_before[-1])
    indptr[mat.shape[0] + 1:] = indptr[mat.shape[0]]
    out = csr_matrix((np.ones(n_kept, dtype=bool), indices, indptr), shape=shape)
    out.has_sorted_indices = True
    return out


_LISTS_CACHE = []   # [(weakrefs of the three matrices, threshold, n_eval_items, result)], most recent first


def eval_lists(train_mat, test_mat, val_mat, rating_threshold, n_eval_items):
    """`_eval_lists` memoised on the IDENTITY of the matrices: an experiment evaluates every model (and both the test
    and the validation pass) on the same split, and the lists depend on nothing else.  (The matrices are the cached
    `Dataset.csr_matrix` objects; a dataset that is modified builds new ones.)"""
    import weakref

    mats = (train_mat, test_mat, val_mat)
    # entries whose matrices have died are dropped (their int64 lists are hundreds of MB at the ML-20M / Netflix shapes)
    _LISTS_CACHE[:] = [e for e in _LISTS_CACHE if all(r is None or r() is not None for r in e[0])]
    for entry in _LISTS_CACHE:
        refs, thr, n_items, nnzs, result = entry
        if thr == rating_threshold and n_items == n_eval_items and nnzs == tuple(None if m is None else m.nnz for m in mats) and all(
                (m is None and r is None) or (r is not None and m is not None and r() is m) for m, r in zip(mats, refs)):
            return result
    result = _eval_lists(train_mat, test_mat, val_mat, rating_threshold, n_eval_items)
    try:
        refs = tuple(None if m is None else weakref.ref(m) for m in mats)
    except TypeError:
        return result
    _LISTS_CACHE.insert(0, (refs, rating_threshold, n_eval_items, tuple(None if m is None else m.nnz for m in mats), result))
    del _LISTS_CACHE[2:]
    return result


def _eval_lists(train_mat, test_mat, val_mat, rating_threshold, n_eval_items):
    """Vectorised form of the per-user mask building of the reference's ranking_eval
    (cornac/eval_methods/base_method.py:176-206), for all test users at once:

      users   users with at least one test positive (ascending),
      gt      their test positives below n_eval_items (sorted)                       -> (gt_ptr, gt_idx)
      excl    (train positives U validation positives) minus test positives, < n_eval_items (sorted) -> (ex_ptr, ex_idx)

    i.e. the candidate set of a user is everything that is not in `excl`.  CSR set algebra instead of three
    numpy set operations per user."""
    mats = [m for m in (train_mat, test_mat, val_mat) if m is not None]
    shape = (max(m.shape[0] for m in mats), max(max(m.shape[1] for m in mats), n_eval_items))
    T = _bool_csr(test_mat, rating_threshold, shape)
    P = _bool_csr(train_mat, rating_threshold, shape)
    if val_mat is not None:
        P = P + _bool_csr(val_mat, rating_threshold, shape)
    users = np.flatnonzero(np.diff(T.indptr) > 0)
    E = (P - P.multiply(T)).tocsr()
    E.eliminate_zeros()
    def cut(M):   # rows of the test users, columns below n_eval_items — without copying when nothing is cut
        if len(users) != M.shape[0]:
            M = M[users]
        if n_eval_items < M.shape[1]:
            M = M[:, :n_eval_items]
        return M.tocsr()

    E = cut(E)
    G = cut(T)
    E.sort_indices()
    G.sort_indices()
    return (users.astype(np.int64), G.indptr.astype(np.int64), G.indices.astype(np.int64), E.indptr.astype(np.int64),
            E.indices.astype(np.int64))


def eval_lists_loop(train_mat, test_mat, val_mat, rating_threshold, n_eval_items, test_users):
    """the same lists user by user with numpy set operations (the reference's formulation); kept as the
    specification eval_lists is tested against"""
    users, gt_pos, excl = [], [], []
    for user_idx in sorted(set(int(u) for u in test_users)):
        tp = _pos_items(test_mat, user_idx, rating_threshold)
        if len(tp) == 0:
            continue
        vp = np.empty(0, np.int64) if val_mat is None else _pos_items(val_mat, user_idx, rating_threshold)
        trp = _pos_items(train_mat, user_idx, rating_threshold)
        ex = np.setdiff1d(np.union1d(vp, trp), tp)
        users.append(user_idx)
        gt_pos.append(np.sort(tp[tp < n_eval_items]))
        excl.append(ex[ex < n_eval_items])
    return users, gt_pos, excl


def ranking_eval(model, metrics, train_set, test_set, val_set=None, rating_threshold=1.0, exclude_unknowns=True,
                 verbose=False, batch_users=16384, batch_users_full=1024, batch_users_topk=65536):
    if len(metrics) == 0:
        return [], []
    max_k = max(m.k for m in metrics)
    need_full = any(m.k <= 0 for m in metrics)
    test_mat, train_mat = test_set.csr_matrix, train_set.csr_matrix
    val_mat = None if val_set is None else val_set.csr_matrix
    n_eval_items = train_set.num_items if exclude_unknowns else test_set.num_items
    user_results = [{} for _ in metrics]

    users, gt_ptr, gt_idx, ex_ptr, ex_idx = eval_lists(train_mat, test_mat, val_mat, rating_threshold, n_eval_items)

    def gt_of(r):
        return gt_idx[gt_ptr[r]:gt_ptr[r + 1]]

    def per_user(r):
        """the reference's flow for one user (base_method.py:176-220) through model.rank()"""
        user_idx, gp = int(users[r]), gt_of(r)
        item_indices = np.setdiff1d(np.arange(n_eval_items), ex_idx[ex_ptr[r]:ex_ptr[r + 1]])
        gt_neg = np.setdiff1d(item_indices, gp)
        # the reference asks for k = max_k (base_method.py:208-210), but its rank() returns ALL candidates with only
        # the first max_k in order, and the metrics over the whole list (k = -1) read past them: hand those the
        # exact full ranking, which agrees with the reference wherever its result does not hinge on that
        # unspecified tail order
        rank_, scores_ = model.rank(user_idx=user_idx, item_indices=item_indices, k=-1 if need_full else max_k)
        for i, mt in enumerate(metrics):
            user_results[i][user_idx] = mt.compute(gt_pos=gp, gt_neg=gt_neg, pd_rank=rank_, pd_scores=scores_,
                                                   item_indices=item_indices)

    # Users the model has no device row for (test-only users of a model trained over num_users, e.g. MF with
    # exclude_unknowns=False: the reference still evaluates them — MF.score falls back to global_mean + i_biases,
    # recom_mf.py:281-286) cannot go through the batched entry points: they take the per-user flow, the rest is
    # re-indexed so that the batched paths below see only users with a row.
    if hasattr(model, "rank_batch") and hasattr(model, "_scorer_rows") and len(users):
        has_row = np.asarray(model._scorer_rows(users)) >= 0
        if not has_row.all():
            for r in np.flatnonzero(~has_row):
                per_user(int(r))
            keep = np.flatnonzero(has_row)

            def take(ptr, idx):
                cnt = np.diff(ptr)[keep]
                newptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
                pos = np.repeat(ptr[keep] - newptr[:-1], cnt) + np.arange(newptr[-1])
                return newptr, idx[pos]

            gt_ptr, gt_idx = take(gt_ptr, gt_idx)
            ex_ptr, ex_idx = take(ex_ptr, ex_idx)
            users = users[keep]

    def batch_hits(items, b0, b1):
        """hits[r, p]: the p-th ranked item of batch row r is one of its test positives — one sorted-key lookup for
        the whole batch (key = batch row * n_items + item) instead of an np.isin per user and metric"""
        n_cols = max(int(n_eval_items), 1)
        rows = np.repeat(np.arange(b1 - b0), np.diff(gt_ptr[b0:b1 + 1]))
        gt_keys = rows * n_cols + gt_idx[gt_ptr[b0]:gt_ptr[b1]]           # ascending: rows, then sorted items
        pred = items.astype(np.int64)
        if len(gt_keys) == 0:
            return np.zeros(pred.shape, bool)
        pred_keys = np.where(pred >= 0, np.arange(b1 - b0)[:, None] * n_cols + pred, -1)
        pos = np.minimum(np.searchsorted(gt_keys, pred_keys), len(gt_keys) - 1)
        return gt_keys[pos] == pred_keys

    # the batched entry points rank over the rows of the model's device item table; they stand in for the per-user flow
    # only when that is exactly the evaluated item range (with exclude_unknowns=False the test set may bring items a
    # model such as MF has no row for — the reference pads those with the minimum score, recommender.py:510-517)
    ranked_items = getattr(model, "batch_num_items", getattr(model, "total_items", n_eval_items))
    batchable = hasattr(model, "rank_batch") and ranked_items == n_eval_items
    full_ok = batchable and need_full and all(
        hasattr(m, "compute_full_batch") if m.k <= 0 else hasattr(m, "compute_batch") for m in metrics)
    if full_ok and hasattr(model, "rank_positions_batch"):
        # the full-list metrics only need to know where each test positive stands among the user's candidates and
        # how long its run of tied scores is: counted on the device, no ranking is produced or copied
        topk_k = max([m.k for m in metrics if m.k > 0], default=0)
        for b0 in range(0, len(users), batch_users):
            b1 = min(b0 + batch_users, len(users))
            ub = [int(u) for u in users[b0:b1]]
            nb = b1 - b0
            ex_ip = (ex_ptr[b0:b1 + 1] - ex_ptr[b0]).astype(np.int64)
            ex_ix = np.ascontiguousarray(ex_idx[ex_ptr[b0]:ex_ptr[b1]], dtype=np.int32)
            gt_ip = (gt_ptr[b0:b1 + 1] - gt_ptr[b0]).astype(np.int64)
            gt_ix = np.ascontiguousarray(gt_idx[gt_ptr[b0]:gt_ptr[b1]], dtype=np.int32)
            _, pos, ge, _ = model.rank_positions_batch(ub, (gt_ip, gt_ix), exclude=(ex_ip, ex_ix))
            n_cand = int(n_eval_items) - np.diff(ex_ip)
            n_gt = np.diff(gt_ip)
            rows = np.repeat(np.arange(nb), n_gt)
            order = np.lexsort((pos, rows))                       # a row's positives in ranked order
            rows, pos, end = rows[order], pos[order].astype(np.int64), ge[order].astype(np.int64) - 1
            starts = gt_ip
            key = rows * int(n_eval_items) + pos
            cum_end = np.searchsorted(key, rows * int(n_eval_items) + end, side="right") - starts[rows]
            runs = (rows, pos, end, cum_end, starts)
            width = max(topk_k, 1)
            hits = np.zeros((nb, width), bool)                    # what the @k metrics see
            head = pos < width
            hits[rows[head], pos[head]] = True
            for i, mt in enumerate(metrics):
                if mt.k <= 0:
                    vals = mt.compute_full_batch(hits, None, n_cand, n_gt, runs=runs)
                elif getattr(mt, "name", "").startswith("NCRR"):
                    vals = mt.compute_batch(hits[:, :topk_k], n_gt, n_pred=n_cand)
                else:
                    vals = mt.compute_batch(hits[:, :topk_k], n_gt)
                user_results[i].update(zip(ub, np.asarray(vals, dtype=float).tolist()))
    elif full_ok:
        # metrics over the full candidate list (AUC, MAP, MRR) batched: full rankings of a block of users from the
        # device (score tile + per-row sort), then positional / tie-aware vectorised forms of the metrics
        topk_k = max([m.k for m in metrics if m.k > 0], default=0)
        marks = np.zeros(max(int(n_eval_items), 1), bool)
        for b0 in range(0, len(users), batch_users_full):
            b1 = min(b0 + batch_users_full, len(users))
            ub = [int(u) for u in users[b0:b1]]
            indptr = (ex_ptr[b0:b1 + 1] - ex_ptr[b0]).astype(np.int64)
            indices = np.ascontiguousarray(ex_idx[ex_ptr[b0]:ex_ptr[b1]], dtype=np.int32)
            items, scores = model.rank_batch(ub, k=-1, exclude=(indptr, indices))
            n_cand = (items >= 0).sum(axis=1)
            n_gt = np.diff(gt_ptr[b0:b1 + 1])
            hits = np.zeros(items.shape, bool)            # one table lookup per row (the lists are a whole catalogue long)
            for r in range(b1 - b0):
                marks[gt_of(b0 + r)] = True
                hits[r, :n_cand[r]] = marks[items[r, :n_cand[r]]]
                marks[gt_of(b0 + r)] = False
            runs = positive_runs(hits, scores, n_cand)
            for i, mt in enumerate(metrics):
                if mt.k <= 0:
                    vals = mt.compute_full_batch(hits, scores, n_cand, n_gt, runs=runs)
                elif getattr(mt, "name", "").startswith("NCRR"):
                    vals = mt.compute_batch(hits[:, :topk_k], n_gt, n_pred=n_cand)
                else:
                    vals = mt.compute_batch(hits[:, :topk_k], n_gt)
                user_results[i].update(zip(ub, np.asarray(vals, dtype=float).tolist()))
    elif need_full or not batchable:
        for r in range(len(users)):
            per_user(r)
    else:
        # the split's exclusion lists go to the device ONCE per (model scorer, split) and stay there across the batches,
        # the metrics passes and the epochs / models evaluated on the split (what bench.py's ranking leg times); only
        # ranked item ids come back
        resident = (hasattr(model, "register_exclusions") and bool(np.all(np.diff(users) > 0)) and
                    model.register_exclusions((id(ex_idx), len(ex_idx), int(n_eval_items), len(users)), users, ex_ptr, ex_idx))
        for b0 in range(0, len(users), batch_users_topk):
            b1 = min(b0 + batch_users_topk, len(users))
            ub = [int(u) for u in users[b0:b1]]
            if resident:
                items = model.rank_batch_resident(ub, k=max_k)
            else:
                indptr = (ex_ptr[b0:b1 + 1] - ex_ptr[b0]).astype(np.int64)  # exclusion CSR of the batch: a slice, no copies per user
                indices = np.ascontiguousarray(ex_idx[ex_ptr[b0]:ex_ptr[b1]], dtype=np.int32)
                items, _ = model.rank_batch(ub, k=max_k, exclude=(indptr, indices))
            hits = batch_hits(items, b0, b1)
            n_gt = np.diff(gt_ptr[b0:b1 + 1])
            slow = []
            for i, mt in enumerate(metrics):
                if hasattr(mt, "compute_batch") and 0 < mt.k <= hits.shape[1]:
                    if getattr(mt, "name", "").startswith("NCRR"):
                        vals = mt.compute_batch(hits, n_gt, n_pred=(items >= 0).sum(axis=1))
                    else:
                        vals = mt.compute_batch(hits, n_gt)
                    user_results[i].update(zip(ub, np.asarray(vals, dtype=float).tolist()))
                else:
                    slow.append((i, mt))
            for r, user_idx in enumerate(ub) if slow else ():
                pd_rank = items[r][items[r] >= 0].astype(np.int64)
                for i, mt in slow:
                    user_results[i][user_idx] = mt.compute(gt_pos=gt_of(b0 + r), gt_neg=None, pd_rank=pd_rank,
                                                           pd_scores=None, item_indices=None)
    avg_results = [sum(ur.values()) / len(ur) for ur in user_results]
    return avg_results, user_results


def scalar_of(v):
    return v.item() if hasattr(v, "item") else v


def rating_eval(model, metrics, test_set, user_based=False, verbose=False):
    """base_method.py:35-105 with one batched prediction kernel for all test ratings."""
    if len(metrics) == 0:
        return [], []
    u_indices, i_indices, r_values = test_set.uir_tuple
    if hasattr(model, "rate_batch"):
        r_preds = model.rate_batch(u_indices, i_indices)
    else:   # a model without the batched kernel: the reference's pair-by-pair loop
        r_preds = np.fromiter((scalar_of(model.rate(int(u), int(i))) for u, i in zip(u_indices, i_indices)),
                              dtype="float", count=len(u_indices))
    avg_results, user_results = [], []

    def scalar(v):
        return v.item() if hasattr(v, "item") else v

    for mt in metrics:
        if user_based:
            per_user = {}
            order = np.argsort(u_indices, kind="stable")  # one sort instead of one mask per user
            su = u_indices[order]
            cuts = np.flatnonzero(np.diff(su)) + 1
            for sel in np.split(order, cuts):
                per_user[int(u_indices[sel[0]])] = scalar(mt.compute(gt_ratings=r_values[sel], pd_ratings=r_preds[sel]))
            user_results.append(per_user)
            avg_results.append(sum(per_user.values()) / len(per_user))
        else:
            user_results.append({})
            avg_results.append(scalar(mt.compute(gt_ratings=r_values, pd_ratings=r_preds)))
    return avg_results, user_results
