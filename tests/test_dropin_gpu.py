"""GPU twin of tests/test_dropin_cpu.py: the reference is absent on the GPU box, so the model-inside-the-evaluator check
runs against a report the REAL reference produced (tests/golden/eval_report.npz, made by make_eval_golden.py with the
reference's own ranking_eval / rating_eval over its seeded BPR / MF).  Here cornac_amd's models are fitted by the HIP
kernels on the same split and evaluated by cornac_amd.eval (device top-k / rank-position kernels)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RANK_METRICS = [("Recall", 5), ("NDCG", 10), ("Precision", 3), ("AUC", None), ("MAP", None), ("MRR", None)]


def test_device_models_reproduce_the_reference_evaluation_report():
    import cornac_amd as ca
    import cornac_amd.eval as ev
    import cornac_amd.metrics as mm

    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_report.npz"))
    rows = [(int(a), int(b), float(c)) for a, b, c in zip(fx["users"], fx["items"], fx["ratings"])]
    n_train = int(fx["n_train"])
    train = ca.Dataset.build(rows[:n_train])
    test = ca.Dataset.build(rows[n_train:], global_uid_map=train.uid_map, global_iid_map=train.iid_map, exclude_unknowns=True)
    kw = dict(k=int(fx["k"]), max_iter=int(fx["epochs"]), learning_rate=float(fx["lr"]), lambda_reg=float(fx["reg"]),
              seed=int(fx["seed"]))
    mk = lambda: [getattr(mm, n)() if k is None else getattr(mm, n)(k=k) for n, k in RANK_METRICS]  # noqa: E731
    for tag, cls in (("bpr", ca.BPR), ("mf", ca.MF)):
        m = cls(**kw).fit(train)
        assert m.effective_mode == "deterministic"
        avg, per_user = ev.ranking_eval(m, mk(), train, test, rating_threshold=float(fx["rating_threshold"]))
        # identical learned parameters (<= 1e-6) -> the report differs only through fp32 summation order at near-ties
        assert np.allclose(avg, fx[tag + "_rank_avg"], atol=3e-3), (tag, avg, fx[tag + "_rank_avg"])
        assert sorted(per_user[0].keys()) == fx[tag + "_rank_users"].tolist()
        # the per-user flow through rank() (what the reference's unmodified ranking_eval calls) gives the same report
        plain = type("PlainModel", (), {"rank": lambda self, **kw2: m.rank(**kw2)})()
        avg2, _ = ev.ranking_eval(plain, mk(), train, test, rating_threshold=float(fx["rating_threshold"]))
        assert np.allclose(avg2, avg, atol=1e-9)
        if tag == "mf":
            avg_r, _ = ev.rating_eval(m, [mm.RMSE(), mm.MAE()], test)
            assert np.allclose(avg_r, fx["mf_rating_avg"], atol=1e-4)
