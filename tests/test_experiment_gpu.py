"""The reference's first example (examples/first_example.py: RatioSplit -> Experiment over MF and BPR with rating and
ranking metrics) end to end on the device, on ML-100K-sized synthetic ratings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_first_example_flow_end_to_end(capsys):
    from cornac_amd import BPR, MF, Experiment, RatioSplit, eval as ev, metrics as mm

    rs = np.random.RandomState(1)
    keys = rs.permutation(943 * 400)[:40000]
    data = [("u%d" % (k // 400), "i%d" % (k % 400), float(rs.randint(1, 6))) for k in keys]
    method = RatioSplit(data, test_size=0.2, rating_threshold=4.0, seed=123)
    models = [MF(k=10, max_iter=25, learning_rate=0.01, lambda_reg=0.02, use_bias=True, seed=123),
              BPR(k=10, max_iter=50, learning_rate=0.01, lambda_reg=0.01, seed=123)]
    metrics = [mm.MAE(), mm.RMSE(), mm.Recall(k=20), mm.Precision(k=20), mm.AUC(), mm.MAP()]
    exp = Experiment(method, models, metrics, user_based=True).run()
    out = capsys.readouterr().out
    assert "TEST:" in out and "MF" in out and "BPR" in out
    mf_res, bpr_res = exp.result
    assert list(mf_res.metric_avg_results) == ["MAE", "RMSE", "AUC", "MAP", "Precision@20", "Recall@20", "Train (s)",
                                               "Test (s)"]
    assert 0.5 < mf_res.metric_avg_results["MAE"] < 2.0 and mf_res.metric_avg_results["RMSE"] >= mf_res.metric_avg_results["MAE"]
    assert 0.3 < bpr_res.metric_avg_results["AUC"] < 1.0
    # the Result rows are exactly what the evaluation loops return for the fitted models
    avg, _ = ev.ranking_eval(models[1], [mm.AUC(), mm.MAP(), mm.Precision(k=20), mm.Recall(k=20)], method.train_set,
                             method.test_set, rating_threshold=4.0, exclude_unknowns=True)
    got = [bpr_res.metric_avg_results[n] for n in ("AUC", "MAP", "Precision@20", "Recall@20")]
    assert np.allclose(avg, got, rtol=0, atol=1e-12)
    (mae,), _ = ev.rating_eval(models[0], [mm.MAE()], method.test_set, user_based=True)
    assert mae == pytest.approx(mf_res.metric_avg_results["MAE"], abs=1e-12)


def test_grid_search_over_bpr_uses_the_batched_entry_points():
    """hyper-parameter search (cornac/hyperopt.py flow) around the device BPR: every grid point trains a clone and is
    scored on the validation set; the searcher then evaluates like its best model, batched entry points included"""
    from cornac_amd import BPR, RatioSplit, metrics as mm
    from cornac_amd.hyperopt import Discrete, GridSearch

    rs = np.random.RandomState(2)
    keys = rs.permutation(600 * 300)[:30000]
    data = [("u%d" % (k // 300), "i%d" % (k % 300), float(rs.randint(1, 6))) for k in keys]
    method = RatioSplit(data, test_size=0.2, val_size=0.1, rating_threshold=1.0, seed=5)
    gs = GridSearch(BPR(k=8, max_iter=20, seed=3), [Discrete("learning_rate", [0.05, 0.001]), Discrete("k", [8, 16])],
                    mm.AUC(), method)
    test_res, val_res = method.evaluate(gs, [mm.AUC(), mm.Recall(k=10)], user_based=True)
    assert set(gs.best_params) == {"k", "learning_rate"} and gs.best_model.k == gs.best_params["k"]
    assert hasattr(gs, "rank_batch") and hasattr(gs, "rank_positions_batch")
    assert val_res.metric_avg_results["AUC"] == pytest.approx(gs.best_score, abs=1e-12)
    assert 0.3 < test_res.metric_avg_results["AUC"] < 1.0
