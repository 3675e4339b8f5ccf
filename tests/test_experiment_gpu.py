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
