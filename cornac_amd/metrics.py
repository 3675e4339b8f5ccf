"""Ranking/rating metrics consumed by the evaluation helpers — mirrors of the reference's `cornac.metrics`
classes (cornac/metrics/ranking.py: NDCG, NCRR, MRR, Precision, Recall, FMeasure, HitRatio, AUC, MAP;
cornac/metrics/rating.py: RMSE, MAE) with the same `compute(...)` keyword interface, so the reference's own
metric objects can be passed to `cornac_amd.eval.ranking_eval` interchangeably.  Each class adds a batched form
(`compute_batch` / `compute_full_batch`) that evaluates all users of a ranked batch at once."""
import numpy as np


class RankingMetric:
    def __init__(self, name, k=-1, higher_better=True):
        self.type = "ranking"
        self.name = name
        self.k = k
        self.higher_better = higher_better


# Batched forms (`compute_batch`) evaluate all users of a ranked batch at once from
#   hits [n_users, K] bool : the p-th ranked item of the user is one of its positives (False beyond the user's list),
#   n_gt [n_users]         : number of positives of the user,
# and return exactly what `compute` returns user by user (tests/test_eval_cpu.py); k must be > 0 and <= K.
def _listed_positives(hits, n_cand, rank_len, runs):
    """(rows, pos, list_len): the positives inside each user's ranked list pd_rank (= the whole candidate list, or
    its first rank_len entries) and that list's length — from `positive_runs`-style `runs` when given, else from
    the dense `hits` matrix.  Used by the k = -1 forms of the @k metrics."""
    if runs is not None:
        rows, pos = runs[0], runs[1]
    else:
        rows, pos = np.nonzero(hits)
    list_len = np.asarray(n_cand) if rank_len is None else np.minimum(n_cand, rank_len)
    inside = pos < list_len[rows]
    return rows[inside], pos[inside], list_len


class _MeasureAtK(RankingMetric):
    def _tp_full(self, hits, n_cand, rank_len, runs):
        rows, _, list_len = _listed_positives(hits, n_cand, rank_len, runs)
        return np.bincount(rows, minlength=len(list_len)).astype(float), list_len

    def _tp(self, gt_pos, pd_rank):
        top = pd_rank[: self.k] if self.k > 0 else pd_rank
        tp = np.sum(np.isin(top, gt_pos))
        return tp, len(gt_pos), (self.k if self.k > 0 else len(top))

    def _tp_batch(self, hits):
        return hits[:, : self.k].sum(axis=1)


class Precision(_MeasureAtK):
    def __init__(self, k=-1):
        super().__init__("Precision@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        tp, _, tp_fp = self._tp(gt_pos, pd_rank)
        return tp / tp_fp

    def compute_batch(self, hits, n_gt):
        return self._tp_batch(hits) / self.k

    def compute_full_batch(self, hits, scores, n_cand, n_gt, rank_len=None, runs=None):   # k = -1
        tp, list_len = self._tp_full(hits, n_cand, rank_len, runs)
        with np.errstate(divide="ignore", invalid="ignore"):
            return tp / list_len


class Recall(_MeasureAtK):
    def __init__(self, k=-1):
        super().__init__("Recall@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        tp, tp_fn, _ = self._tp(gt_pos, pd_rank)
        return tp / tp_fn

    def compute_batch(self, hits, n_gt):
        with np.errstate(divide="ignore", invalid="ignore"):
            return self._tp_batch(hits) / n_gt

    def compute_full_batch(self, hits, scores, n_cand, n_gt, rank_len=None, runs=None):   # k = -1
        tp, _ = self._tp_full(hits, n_cand, rank_len, runs)
        with np.errstate(divide="ignore", invalid="ignore"):
            return tp / n_gt


class HitRatio(_MeasureAtK):
    def __init__(self, k=-1):
        super().__init__("HitRatio@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        return 1.0 if self._tp(gt_pos, pd_rank)[0] > 0 else 0.0

    def compute_batch(self, hits, n_gt):
        return (self._tp_batch(hits) > 0).astype(float)

    def compute_full_batch(self, hits, scores, n_cand, n_gt, rank_len=None, runs=None):   # k = -1
        return (self._tp_full(hits, n_cand, rank_len, runs)[0] > 0).astype(float)


class NDCG(RankingMetric):
    def __init__(self, k=-1):
        super().__init__("NDCG@{}".format(k), k)

    @staticmethod
    def dcg_score(gt_pos, pd_rank, k=-1):
        top = pd_rank[:k] if k > 0 else pd_rank
        rel = np.isin(top, gt_pos).astype(int)
        return np.sum((2 ** rel - 1) / np.log2(np.arange(len(rel)) + 2))

    def compute(self, gt_pos, pd_rank, **kwargs):
        return self.dcg_score(gt_pos, pd_rank, self.k) / self.dcg_score(gt_pos, gt_pos, self.k)

    def compute_batch(self, hits, n_gt):
        disc = 1.0 / np.log2(np.arange(self.k) + 2)
        dcg = (hits[:, : self.k] * disc).sum(axis=1)
        ideal = np.concatenate([[0.0], np.cumsum(disc)])[np.minimum(n_gt, self.k)]
        with np.errstate(divide="ignore", invalid="ignore"):
            return dcg / ideal

    def compute_full_batch(self, hits, scores, n_cand, n_gt, rank_len=None, runs=None):   # k = -1
        rows, pos, list_len = _listed_positives(hits, n_cand, rank_len, runs)
        dcg = np.bincount(rows, weights=1.0 / np.log2(pos + 2.0), minlength=len(list_len))
        top = int(np.max(n_gt, initial=0))
        ideal = np.concatenate([[0.0], np.cumsum(1.0 / np.log2(np.arange(top) + 2.0))])[n_gt]   # the positives ranked first
        with np.errstate(divide="ignore", invalid="ignore"):
            return dcg / ideal


class FMeasure(_MeasureAtK):
    def __init__(self, k=-1):
        super().__init__("F1@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        tp, tp_fn, tp_fp = self._tp(gt_pos, pd_rank)
        prec, rec = tp / tp_fp, tp / tp_fn
        return 2 * (prec * rec) / (prec + rec) if (prec + rec) > 0 else 0

    def compute_batch(self, hits, n_gt):
        tp = self._tp_batch(hits)
        with np.errstate(divide="ignore", invalid="ignore"):
            prec, rec = tp / self.k, tp / n_gt
            f1 = 2 * (prec * rec) / (prec + rec)
        return np.where(prec + rec > 0, f1, 0.0)

    def compute_full_batch(self, hits, scores, n_cand, n_gt, rank_len=None, runs=None):   # k = -1
        tp, list_len = self._tp_full(hits, n_cand, rank_len, runs)
        with np.errstate(divide="ignore", invalid="ignore"):
            prec, rec = tp / list_len, tp / n_gt
            f1 = 2 * (prec * rec) / (prec + rec)
        return np.where(prec + rec > 0, f1, 0.0)


class NCRR(RankingMetric):
    """normalised cumulative reciprocal rank (cornac/metrics/ranking.py:122-177)"""

    def __init__(self, k=-1):
        super().__init__("NCRR@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        top = pd_rank[: self.k] if self.k > 0 else pd_rank
        where = np.flatnonzero(np.isin(top, gt_pos))
        if len(where) == 0:
            return 0.0
        ideal = min(len(gt_pos), len(top))
        return np.sum(1.0 / (where + 1)) / np.sum(1.0 / (np.arange(ideal) + 1))

    def compute_batch(self, hits, n_gt, n_pred=None):
        """n_pred: length of each user's ranked list (the ideal run is min(n_gt, min(n_pred, k)))"""
        inv = 1.0 / (np.arange(self.k) + 1)
        crr = (hits[:, : self.k] * inv).sum(axis=1)
        run = np.minimum(n_gt, self.k if n_pred is None else np.minimum(n_pred, self.k))
        ideal = np.concatenate([[0.0], np.cumsum(inv)])[run]
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(crr > 0, crr / ideal, 0.0)

    def compute_full_batch(self, hits, scores, n_cand, n_gt, rank_len=None, runs=None):   # k = -1
        rows, pos, list_len = _listed_positives(hits, n_cand, rank_len, runs)
        crr = np.bincount(rows, weights=1.0 / (pos + 1.0), minlength=len(list_len))
        run = np.minimum(n_gt, list_len)
        ideal = np.concatenate([[0.0], np.cumsum(1.0 / (np.arange(int(np.max(run, initial=0))) + 1.0))])[run]
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(crr > 0, crr / ideal, 0.0)


# Metrics over the FULL ranked candidate list.  Their batched forms take, per user row,
#   hits   [n, L] bool   : the p-th ranked candidate is a positive (False beyond n_cand),
#   scores [n, L] float  : the candidates' scores in ranked (descending) order,
#   n_cand [n], n_gt [n] : number of candidates / of positives among them,
#   rank_len             : length of the ranked prefix the per-user flow would pass as pd_rank (None = all),
#   runs                 : `positive_runs(hits, scores, n_cand)` when the caller shares it between metrics,
# and treat tied scores exactly like the score-based per-user definitions (strict ">" in AUC, "max" ranks in MAP).
# Only the positives are visited (a few per row): one pass over `hits` finds them, a binary search in the row's
# sorted scores finds the end of each one's run of tied scores.
def positive_runs(hits, scores, n_cand):
    """(rows, pos, end, cum_end, starts): for every positive, its row and position, the last position of its run of
    equal scores, the number of the row's positives up to that position; starts[r]:starts[r+1] = positives of row r"""
    n, L = hits.shape
    rows, pos = np.nonzero(hits)                     # row-major: positions ascending within a row
    starts = np.searchsorted(rows, np.arange(n + 1))
    end = np.empty_like(pos)
    for r in np.flatnonzero(np.diff(starts)):
        a, b = starts[r], starts[r + 1]
        row = scores[r, : n_cand[r]]
        end[a:b] = len(row) - 1 - np.searchsorted(row[::-1], row[pos[a:b]], side="left")
    key = rows * L + pos
    cum_end = np.searchsorted(key, rows * L + end, side="right") - starts[rows]
    return rows, pos, end, cum_end, starts


class MRR(RankingMetric):
    def __init__(self):
        super().__init__("MRR")

    def compute(self, gt_pos, pd_rank, **kwargs):
        where = np.flatnonzero(np.isin(pd_rank, gt_pos))
        if len(where) == 0:
            raise ValueError("No matched between ground-truth items and recommendations")
        return 1.0 / (where[0] + 1)

    def compute_full_batch(self, hits, scores, n_cand, n_gt, rank_len=None, runs=None):
        hits = hits if rank_len is None else hits[:, :rank_len]
        if runs is not None and rank_len is None:
            _, pos, _, _, starts = runs
            if (np.diff(starts) == 0).any():
                raise ValueError("No matched between ground-truth items and recommendations")
            return 1.0 / (pos[starts[:-1]] + 1)
        if not hits.any(axis=1).all():
            raise ValueError("No matched between ground-truth items and recommendations")
        return 1.0 / (hits.argmax(axis=1) + 1)


class AUC(RankingMetric):
    """fraction of (positive, negative) candidate pairs ranked correctly (cornac/metrics/ranking.py:428-485)"""

    def __init__(self):
        super().__init__("AUC")

    def compute(self, item_indices, pd_scores, gt_pos, gt_neg=None, **kwargs):
        pos_mask = np.isin(item_indices, gt_pos)
        neg_mask = np.logical_not(pos_mask) if gt_neg is None else np.isin(item_indices, gt_neg)
        pos, neg = pd_scores[pos_mask], pd_scores[neg_mask]
        return (pos[:, None] > neg[None, :]).sum() / (len(pos) * len(neg))

    def compute_full_batch(self, hits, scores, n_cand, n_gt, rank_len=None, runs=None):
        rows, _, end, cum_end, _ = positive_runs(hits, scores, n_cand) if runs is None else runs
        n_neg = n_cand - n_gt
        # negatives scored strictly below a positive = all negatives - negatives up to the end of its run of ties
        below = n_neg[rows] - ((end + 1) - cum_end)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.bincount(rows, weights=below, minlength=len(n_gt)) / (n_gt * n_neg)


class MAP(RankingMetric):
    """mean average precision with "max" ranks for tied scores (cornac/metrics/ranking.py:488-527)"""

    def __init__(self):
        super().__init__("MAP")

    def compute(self, item_indices, pd_scores, gt_pos, **kwargs):
        from scipy.stats import rankdata

        relevant = np.isin(item_indices, gt_pos)
        rank = rankdata(-pd_scores, "max")[relevant]
        among = rankdata(-pd_scores[relevant], "max")
        return (among / rank).mean()

    def compute_full_batch(self, hits, scores, n_cand, n_gt, rank_len=None, runs=None):
        rows, _, end, cum_end, _ = positive_runs(hits, scores, n_cand) if runs is None else runs
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.bincount(rows, weights=cum_end / (end + 1.0), minlength=len(n_gt)) / n_gt


class _RatingMetric:
    """rating metrics take optional per-rating weights like the reference's (cornac/metrics/rating.py:40-140)"""
    type = "rating"
    higher_better = False

    @staticmethod
    def _diff(gt_ratings, pd_ratings):
        return np.asarray(gt_ratings, float) - np.asarray(pd_ratings, float)


class MSE(_RatingMetric):
    name = "MSE"

    def compute(self, gt_ratings, pd_ratings, weights=None, **kwargs):
        d = self._diff(gt_ratings, pd_ratings)
        return float(np.average(d * d, axis=0, weights=weights))


class RMSE(_RatingMetric):
    name = "RMSE"

    def compute(self, gt_ratings, pd_ratings, weights=None, **kwargs):
        d = self._diff(gt_ratings, pd_ratings)
        return float(np.sqrt(np.average(d * d, axis=0, weights=weights)))


class MAE(_RatingMetric):
    name = "MAE"

    def compute(self, gt_ratings, pd_ratings, weights=None, **kwargs):
        return float(np.average(np.abs(self._diff(gt_ratings, pd_ratings)), axis=0, weights=weights))
