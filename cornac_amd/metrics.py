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


class _MeasureAtK(RankingMetric):
    def _tp(self, gt_pos, pd_rank):
        top = pd_rank[: self.k] if self.k > 0 else pd_rank
        tp = np.sum(np.isin(top, gt_pos))
        return tp, len(gt_pos), (self.k if self.k > 0 else len(top))


class Precision(_MeasureAtK):
    def __init__(self, k=-1):
        super().__init__("Precision@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        tp, _, tp_fp = self._tp(gt_pos, pd_rank)
        return tp / tp_fp


class Recall(_MeasureAtK):
    def __init__(self, k=-1):
        super().__init__("Recall@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        tp, tp_fn, _ = self._tp(gt_pos, pd_rank)
        return tp / tp_fn


class HitRatio(_MeasureAtK):
    def __init__(self, k=-1):
        super().__init__("HitRatio@{}".format(k), k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        return 1.0 if self._tp(gt_pos, pd_rank)[0] > 0 else 0.0


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
