"""Ranking/rating metrics consumed by the evaluation helpers — minimal mirrors of the reference's
`cornac.metrics` classes for the @k family (cornac/metrics/ranking.py:226-430) and RMSE/MAE
(cornac/metrics/rating.py), with the same `compute(...)` keyword interface, so the reference's own
metric objects can be passed to `cornac_amd.eval.ranking_eval` interchangeably."""
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
class _MeasureAtK(RankingMetric):
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


class Recall(_MeasureAtK):
    def __init__(self, k=-1):
        super().__init__("Recall@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        tp, tp_fn, _ = self._tp(gt_pos, pd_rank)
        return tp / tp_fn

    def compute_batch(self, hits, n_gt):
        with np.errstate(divide="ignore", invalid="ignore"):
            return self._tp_batch(hits) / n_gt


class HitRatio(_MeasureAtK):
    def __init__(self, k=-1):
        super().__init__("HitRatio@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        return 1.0 if self._tp(gt_pos, pd_rank)[0] > 0 else 0.0

    def compute_batch(self, hits, n_gt):
        return (self._tp_batch(hits) > 0).astype(float)


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


class RMSE:
    type, name = "rating", "RMSE"

    @staticmethod
    def compute(gt_ratings, pd_ratings, **kwargs):
        d = np.asarray(gt_ratings, float) - np.asarray(pd_ratings, float)
        return float(np.sqrt(np.mean(d * d)))


class MAE:
    type, name = "rating", "MAE"

    @staticmethod
    def compute(gt_ratings, pd_ratings, **kwargs):
        return float(np.mean(np.abs(np.asarray(gt_ratings, float) - np.asarray(pd_ratings, float))))
