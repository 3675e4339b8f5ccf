"""Known-answer values the reference's own metric tests hold (tests/cornac/metrics/test_ranking.py:45-275,
tests/cornac/metrics/test_rating.py:35-62), applied to the mirrors in cornac_amd/metrics.py and — where a batched
form exists — to that form as well.  Needs neither the reference tree nor a GPU."""
import numpy as np
import pytest

from cornac_amd import metrics as mm

A = np.asarray

# (metric, gt_pos, pd_rank, expected) — test_ranking.py: ndcg :45-67, ncrr :69-104, mrr :106-126, hit ratio :147-173,
# precision :175-196, recall :198-219, f-measure :221-247
RANK_CASES = [
    (mm.NDCG(), [0], [0], 1), (mm.NDCG(), [0, 2], [0, 2, 1], 1),
    (mm.NCRR(), [0], [0], 1), (mm.NCRR(), [0, 2], [0, 2, 1], 1),
    (mm.NCRR(), [0, 2], [1, 2, 0], (1 / 3 + 1 / 2) / (1 + 1 / 2)),
    (mm.NCRR(k=2), [2], [1, 2, 0], 0.5), (mm.NCRR(k=2), [2], [4, 1, 2], 0.0),
    (mm.NCRR(k=2), [0, 1, 2], [5, 1, 6], 1.0 / 3.0), (mm.NCRR(k=3), [0, 1], [5, 1, 6, 8], 1.0 / 3.0),
    (mm.MRR(), [0], [0], 1), (mm.MRR(), [0, 2], [0, 2, 1], 1), (mm.MRR(), [0, 2], [1, 2, 0], 1 / 2),
    (mm.HitRatio(), [0], [0], 1), (mm.HitRatio(), [0, 1], [0, 2], 1), (mm.HitRatio(), [0, 2], [0, 2, 1], 1),
    (mm.HitRatio(), [2], [1, 2, 0], 1), (mm.HitRatio(k=2), [0], [1, 2, 0], 0), (mm.HitRatio(k=2), [2], [1, 2, 0], 1),
    (mm.Precision(), [0], [0], 1), (mm.Precision(), [0, 2], [0, 2, 1], 2 / 3), (mm.Precision(), [2], [1, 2, 0], 1 / 3),
    (mm.Precision(k=2), [2], [1, 2, 0], 0.5),
    (mm.Recall(), [0], [0], 1), (mm.Recall(), [0, 2], [0, 2, 1], 1), (mm.Recall(), [2], [1, 2, 0], 1),
    (mm.Recall(k=2), [2], [1, 2, 0], 1),
    (mm.FMeasure(), [0], [0], 1), (mm.FMeasure(), [0, 2], [0, 2, 1], 4 / 5), (mm.FMeasure(), [2], [1, 2, 0], 1 / 2),
    (mm.FMeasure(k=2), [2], [1, 2, 0], 2 / 3), (mm.FMeasure(k=2), [0], [1, 2], 0),
]


@pytest.mark.parametrize("case", range(len(RANK_CASES)))
def test_rank_based_metrics_reproduce_the_reference_tests_values(case):
    mt, gt, rank, want = RANK_CASES[case]
    gt, rank = A(gt), A(rank)
    assert mt.compute(gt_pos=gt, pd_rank=rank) == want
    # the same through the batched form: a one-user batch whose ranked list is `rank`
    hits = np.isin(rank, gt)[None, :]
    n_gt, n_cand = A([len(gt)]), A([len(rank)])
    if mt.k > 0 and mt.k <= len(rank):
        got = mt.compute_batch(hits, n_gt, n_pred=n_cand) if isinstance(mt, mm.NCRR) else mt.compute_batch(hits, n_gt)
        assert float(got[0]) == pytest.approx(want, rel=1e-15)
    elif mt.k <= 0:
        got = mt.compute_full_batch(hits, -np.arange(len(rank), dtype=np.float32)[None, :], n_cand, n_gt)
        assert float(got[0]) == pytest.approx(want, rel=1e-15)


def test_names_types_and_the_rounded_ndcg_value():
    # test_ranking.py:33-43, 47-48, 61-67
    assert (mm.NDCG().type, mm.NDCG().name, mm.NDCG().k) == ("ranking", "NDCG@-1", -1)
    assert [m.name for m in (mm.NCRR(), mm.MRR(), mm.HitRatio(), mm.Precision(), mm.Recall(), mm.FMeasure(), mm.AUC(),
                             mm.MAP())] == ["NCRR@-1", "MRR", "HitRatio@-1", "Precision@-1", "Recall@-1", "F1@-1",
                                            "AUC", "MAP"]
    assert float("{:.2f}".format(mm.NDCG(k=2).compute(A([2]), A([1, 2, 0])))) == 0.63
    with pytest.raises(ValueError):      # :121-126 no match between the positives and the list
        mm.MRR().compute(A([0, 2]), A([1]))


def test_score_based_metrics_reproduce_the_reference_tests_values():
    # AUC test_ranking.py:249-270
    sc = A([0.1, 0.4, 0.35, 0.8])
    assert mm.AUC().compute(np.arange(4), sc, A([2, 3])) == 0.75
    assert mm.AUC().compute(np.arange(4), sc, A([1, 3])) == 1.0
    assert mm.AUC().compute(np.arange(4), sc, A([2]), A([1, 1, 0, 0])) == 0.5
    # MAP :272-291
    assert mm.MAP().compute(np.arange(3), A([0.75, 0.5, 1]), A([0])) == 0.5
    assert mm.MAP().compute(np.arange(3), A([1, 0.2, 0.1]), A([2])) == 1 / 3
    assert mm.MAP().compute(np.arange(10), np.linspace(0.0, 1.0, 10)[::-1], A([1, 3, 5])) == 0.5
    # the same lists through the batched forms (ranked order = descending score)
    for items, scores, gt, auc, ap in ((np.arange(4), sc, [2, 3], 0.75, None), (np.arange(4), sc, [1, 3], 1.0, None),
                                       (np.arange(3), A([0.75, 0.5, 1]), [0], None, 0.5),
                                       (np.arange(3), A([1, 0.2, 0.1]), [2], None, 1 / 3),
                                       (np.arange(10), np.linspace(0.0, 1.0, 10)[::-1], [1, 3, 5], None, 0.5)):
        order = np.argsort(-scores, kind="stable")
        hits = np.isin(items[order], gt)[None, :]
        args = (hits, scores[order][None, :].astype(np.float64), A([len(items)]), A([len(gt)]))
        if auc is not None:
            assert float(mm.AUC().compute_full_batch(*args)[0]) == auc
        if ap is not None:
            assert float(mm.MAP().compute_full_batch(*args)[0]) == pytest.approx(ap, rel=1e-15)


def test_rating_metrics_reproduce_the_reference_tests_values():
    # test_rating.py:35-62 (third case of each: weights [1, 3])
    for mt, plain, weighted in ((mm.MAE(), 1, 2), (mm.MSE(), 1, 4), (mm.RMSE(), 1, 2)):
        assert mt.type == "rating"
        assert mt.compute(A([0]), A([0])) == 0
        assert mt.compute(A([0, 1]), A([1, 0])) == plain
        assert mt.compute(A([0, 1]), A([2, 3]), A([1, 3])) == weighted
