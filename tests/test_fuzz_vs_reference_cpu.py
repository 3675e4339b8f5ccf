"""Differential fuzz against the LIVE reference (this container only) for the layers around the kernels:
degenerate data through the host model classes (oracle-backed device doubles), `rating_eval` with unknown users and
items, and the split protocols.  Each case compares results or, where both sides fail, the exception type."""
import importlib
import warnings

import numpy as np
import pytest

import fake_device
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference not available / oracle/_ref not built")


def _outcome(fn):
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return fn()
    except Exception as e:   # noqa: BLE001 - the exception type is compared
        return type(e).__name__


def test_degenerate_data_through_the_host_classes(monkeypatch):
    """1-4 users and items, down to a single item or a single interaction (the boost sampler's empty range, all draws
    skipped): factors after 3 seeded epochs equal the reference's"""
    from cornac_amd import BPR, MF, WBPR, Dataset

    fake_device.install(monkeypatch)
    ns = ref_loader.load()
    for seed in range(60):
        rs = np.random.RandomState(seed)
        nu, ni = rs.randint(1, 5), rs.randint(1, 5)
        keys = rs.permutation(nu * ni)[: rs.randint(1, nu * ni + 1)]
        data = [("u%d" % (k // ni), "i%d" % (k % ni), float(rs.randint(1, 6))) for k in keys]
        rd, md = ns.Dataset.from_uir(data, seed=1), Dataset.from_uir(data, seed=1)
        for R, M, kw in ((ns.BPR, BPR, dict(k=int(rs.randint(1, 4)), max_iter=3, seed=seed, learning_rate=0.1)),
                         (ns.WBPR, WBPR, dict(k=2, max_iter=3, seed=seed, learning_rate=0.1)),
                         (ns.MF, MF, dict(k=2, max_iter=3, seed=seed))):
            r, m = _outcome(lambda: R(**kw).fit(rd)), _outcome(lambda: M(**kw).fit(md))
            if isinstance(r, str) or isinstance(m, str):
                assert r == m, (seed, R.__name__, r, m)
                continue
            for name in ("u_factors", "i_factors", "i_biases"):
                assert np.abs(getattr(r, name) - getattr(m, name)).max() < 1e-5, (seed, R.__name__, name)


def test_rating_eval_with_unknown_users_and_items(monkeypatch):
    import cornac_amd.eval as ev
    import cornac_amd.metrics as mm
    from cornac_amd import BPR, MF, Dataset

    fake_device.install(monkeypatch)
    ns = ref_loader.load()
    rm, ref_eval = ns.metrics, ns.eval_methods.base_method.rating_eval
    compared = 0
    for seed in range(16):
        rs = np.random.RandomState(seed)
        nu, ni = rs.randint(15, 60), rs.randint(12, 40)
        keys = rs.permutation(nu * ni)[: min(rs.randint(100, 400), nu * ni - 1)]
        data = [("u%d" % (k // ni), "i%d" % (k % ni), float(rs.randint(1, 6))) for k in keys]
        a, drop = int(len(data) * 0.7), bool(rs.randint(2))
        rtrain, mtrain = ns.Dataset.build(data[:a]), Dataset.build(data[:a])
        rtest = _outcome(lambda: ns.Dataset.build(data[a:], global_uid_map=rtrain.uid_map, global_iid_map=rtrain.iid_map,
                                                  exclude_unknowns=drop))
        if isinstance(rtest, str):
            continue
        mtest = Dataset.build(data[a:], global_uid_map=mtrain.uid_map, global_iid_map=mtrain.iid_map, exclude_unknowns=drop)
        for R, M, kw in ((ns.MF, MF, dict(k=4, max_iter=8, seed=1, use_bias=bool(rs.randint(2)))),
                         (ns.BPR, BPR, dict(k=4, max_iter=5, seed=2))):
            r, m = R(**kw).fit(rtrain), M(**kw).fit(mtrain)
            for user_based in (True, False):
                ra, ru = ref_eval(r, [rm.MAE(), rm.RMSE(), rm.MSE()], rtest, user_based=user_based)
                ma, mu = ev.rating_eval(m, [mm.MAE(), mm.RMSE(), mm.MSE()], mtest, user_based=user_based)
                assert np.allclose(ra, ma, rtol=1e-5, atol=1e-6), (seed, R.__name__, user_based, drop)
                assert all(x.keys() == y.keys() for x, y in zip(ru, mu))
                compared += 1
    assert compared >= 40


def test_split_protocols_on_random_data():
    from cornac_amd import CrossValidation, RatioSplit, StratifiedSplit

    ref_loader.load()
    em = importlib.import_module("cornac.eval_methods")

    def same(a, b):
        if a is None or b is None:
            return a is b
        return (all(np.array_equal(x, y) for x, y in zip(a.uir_tuple, b.uir_tuple))
                and list(a.uid_map.items()) == list(b.uid_map.items()) and list(a.iid_map.items()) == list(b.iid_map.items())
                and ((a.timestamps is None and b.timestamps is None) or np.array_equal(a.timestamps, b.timestamps)))

    compared = 0
    for seed in range(40):
        rs = np.random.RandomState(seed)
        nu, ni = rs.randint(5, 40), rs.randint(5, 30)
        keys = rs.randint(0, nu * ni, rs.randint(30, min(600, nu * ni)))           # duplicated pairs allowed
        uirt = [("u%d" % (k // ni), "i%d" % (k % ni), float(rs.randint(1, 6)), int(rs.randint(0, 1000))) for k in keys]
        uir = [t[:3] for t in uirt]
        cases = [(em.RatioSplit, RatioSplit, uir, dict(test_size=rs.choice([0.1, 0.2, 0.35, 5, 12]), val_size=rs.choice([0.0, 0.1, 3]),
                                                        seed=seed, exclude_unknowns=bool(rs.randint(2)))),
                 (em.StratifiedSplit, StratifiedSplit, uirt, dict(group_by=str(rs.choice(["user", "item"])), chrono=bool(rs.randint(2)),
                                                                  test_size=float(rs.choice([0.2, 0.3])), val_size=float(rs.choice([0.0, 0.2])),
                                                                  seed=seed, exclude_unknowns=bool(rs.randint(2))))]
        for R, M, d, kw in cases:
            r, m = _outcome(lambda: R(d, **kw)), _outcome(lambda: M(d, **kw))
            if isinstance(r, str) or isinstance(m, str):
                assert r == m, (seed, R.__name__, kw, r, m)
                continue
            assert all(same(getattr(r, p), getattr(m, p)) for p in ("train_set", "test_set", "val_set")), (seed, R.__name__, kw)
            compared += 1
        folds = int(rs.randint(2, 6))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert np.array_equal(em.CrossValidation(uir, n_folds=folds, seed=seed)._partition,
                                  CrossValidation(uir, n_folds=folds, seed=seed)._partition)
    assert compared >= 30


def test_init_params_dtypes_and_shapes_through_the_host_classes(monkeypatch):
    """init_params fuzz for BPR / WBPR: float32 or float64 tables (the reference's fused-type `_fit_sgd`), any subset of
    {U, V, Bi} given, use_bias on or off, a dtype mix now and then — same learned tables (dtype included), or the same
    exception type on both sides"""
    from cornac_amd import BPR, WBPR, Dataset

    fake_device.install(monkeypatch)
    ns = ref_loader.load()
    outcomes = {"f32": 0, "f64": 0, "error": 0}
    for seed in range(48):
        rs = np.random.RandomState(1000 + seed)
        nu, ni, k = rs.randint(8, 30), rs.randint(6, 25), int(rs.randint(1, 7))
        keys = rs.permutation(nu * ni)[: rs.randint(ni, nu * ni // 2 + ni)]
        data = [("u%d" % (q // ni), "i%d" % (q % ni), float(rs.randint(1, 6))) for q in keys]
        rd, md = ns.Dataset.from_uir(data, seed=1), Dataset.from_uir(data, seed=1)
        dt = [np.float32, np.float64][seed % 2]
        tables = {"U": ((rs.rand(rd.num_users, k) - 0.5) / k).astype(dt), "V": ((rs.rand(rd.num_items, k) - 0.5) / k).astype(dt),
                  "Bi": (0.01 * rs.randn(rd.num_items)).astype(dt)}
        given = [n for n in ("U", "V", "Bi") if rs.rand() < 0.8]
        if seed % 7 == 3 and "V" in given:
            tables["V"] = tables["V"].astype(np.float64 if dt == np.float32 else np.float32)   # a mix
        kw = dict(k=k, max_iter=3, seed=seed, learning_rate=0.05, use_bias=bool(seed % 5))
        for R, M in ((ns.BPR, BPR), (ns.WBPR, WBPR)):
            r = _outcome(lambda: R(init_params={n: tables[n].copy() for n in given}, **kw).fit(rd))
            m = _outcome(lambda: M(init_params={n: tables[n].copy() for n in given}, **kw).fit(md))
            if isinstance(r, str) or isinstance(m, str):
                assert r == m, (seed, R.__name__, given, r, m)
                outcomes["error"] += 1
                continue
            for name in ("u_factors", "i_factors", "i_biases"):
                a, b = getattr(r, name), getattr(m, name)
                assert a.dtype == b.dtype, (seed, R.__name__, name, a.dtype, b.dtype)
                assert np.abs(a - b).max() < (1e-12 if a.dtype == np.float64 else 1e-5), (seed, R.__name__, name)
            outcomes["f64" if r.u_factors.dtype == np.float64 else "f32"] += 1
            s_r, s_m = r.score(0), m.score(0)
            assert np.abs(s_r - s_m).max() < (1e-12 if s_r.dtype == np.float64 and s_m.dtype == np.float64 else 1e-5)
    assert min(outcomes.values()) > 0, outcomes
