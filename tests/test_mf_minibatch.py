"""MF minibatch/optimiser path (reference: MF(backend="pytorch"), backend_pt.py): oracle vs the real
reference's goldens on CPU, HIP path vs oracle and goldens on the GPU."""
import numpy as np
import pytest

from conftest import golden_dataset, load_golden
from oracle import mf_minibatch_oracle

OPTS = ("sgd", "adam", "rmsprop", "adagrad")


def _setup(fx, use_bias):
    ds = golden_dataset(fx)
    ds.reset()
    k, seed = int(fx["k"]), int(fx["seed"])
    rng = np.random.RandomState(seed)  # recom_mf.py:138-156
    U = rng.normal(0.0, 0.01, (ds.num_users, k)).astype(np.float32)
    V = rng.normal(0.0, 0.01, (ds.num_items, k)).astype(np.float32)
    mu = np.float32(ds.global_mean if use_bias else 0.0)
    rid, cid, val = ds.uir_tuple
    batches = []
    for _ in range(int(fx["epochs"])):
        batches += list(ds.idx_iter(len(val), int(fx["batch_size"]), shuffle=True))
    return ds, U, V, mu, rid, cid, val.astype(np.float32), batches


@pytest.mark.parametrize("use_bias", [True, False])
@pytest.mark.parametrize("opt", OPTS)
def test_oracle_matches_reference_golden(opt, use_bias):
    fx = load_golden("mf_minibatch")
    ds, U, V, mu, rid, cid, val, batches = _setup(fx, use_bias)
    Uo, Vo, Buo, Bio, _ = mf_minibatch_oracle.fit(U, V, np.zeros(ds.num_users), np.zeros(ds.num_items), mu, rid, cid, val,
                                                  batches, opt, float(fx["lr"]), float(fx["reg"]), use_bias)
    tag = opt + ("" if use_bias else "_nobias")
    for got, name in ((Uo, "_U"), (Vo, "_V"), (Buo, "_Bu"), (Bio, "_Bi")):
        assert np.abs(got - fx[tag + name]).max() <= 2e-6, (tag + name, np.abs(got - fx[tag + name]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("use_bias", [True, False])
@pytest.mark.parametrize("opt", OPTS)
def test_hip_minibatch_matches_reference_golden_and_oracle(opt, use_bias):
    from cornac_amd import MF

    fx = load_golden("mf_minibatch")
    ds = golden_dataset(fx)
    m = MF(k=int(fx["k"]), backend="hip-minibatch", optimizer=opt, max_iter=int(fx["epochs"]),
           batch_size=int(fx["batch_size"]), learning_rate=float(fx["lr"]), lambda_reg=float(fx["reg"]),
           use_bias=use_bias, seed=int(fx["seed"])).fit(ds)
    tag = opt + ("" if use_bias else "_nobias")
    for got, name in ((m.u_factors, "_U"), (m.i_factors, "_V"), (m.u_biases, "_Bu"), (m.i_biases, "_Bi")):
        err = np.abs(np.asarray(got) - fx[tag + name]).max()
        assert err <= 1e-4, (tag + name, err)
    assert m.loss_history[-1] < m.loss_history[0]


@pytest.mark.gpu
def test_hip_minibatch_ragged_duplicates_and_errors():
    from cornac_amd import MF, _lib

    rs = np.random.RandomState(3)
    nu, ni, k, n = 40, 30, 5, 333
    rid, cid = rs.randint(0, nu, n).astype(np.int64), rs.randint(0, ni, n).astype(np.int64)  # duplicates inside batches
    val = rs.randint(1, 6, n).astype(np.float32)
    U, V = rs.normal(0, .1, (nu, k)).astype(np.float32), rs.normal(0, .1, (ni, k)).astype(np.float32)
    Bu, Bi = rs.normal(0, .1, nu).astype(np.float32), rs.normal(0, .1, ni).astype(np.float32)
    order = rs.permutation(n)
    batches = [order[s:s + 50] for s in range(0, n, 50)]  # last batch ragged (33)
    for opt in OPTS:
        tr = _lib.MfTrainer(rid, cid, val, nu, ni, k)
        tr.set_factors(U, V, Bu, Bi)
        sse = tr.fit_minibatch(order, 50, opt, 0.01, 0.05, 3.0, True)
        sse += tr.fit_minibatch(order, 50, opt, 0.01, 0.05, 3.0, True)  # optimiser state persists across calls
        got = tr.get_factors()
        with pytest.raises(_lib.HipError):
            tr.fit_minibatch(order, 50, OPTS[(OPTS.index(opt) + 1) % 4], 0.01, 0.05, 3.0, True)
        with pytest.raises(_lib.HipError):
            tr.fit_minibatch(np.array([n]), 50, opt, 0.01, 0.05, 3.0, True)
        tr.close()
        want = mf_minibatch_oracle.fit(U, V, Bu, Bi, 3.0, rid, cid, val, batches + batches, opt, 0.01, 0.05, True)
        for a, b in zip(got, want[:4]):
            assert np.abs(a - b).max() <= 2e-5, (opt, np.abs(a - b).max())
        assert abs(sse - sum(want[4])) <= 1e-4 * sum(want[4])
    with pytest.raises(ValueError, match="dropout probability"):
        MF(backend="hip-minibatch", dropout=1.1).fit(golden_dataset(load_golden("tiny")))
    with pytest.raises(KeyError):
        MF(backend="hip-minibatch", optimizer="lbfgs").fit(golden_dataset(load_golden("tiny")))


DROPOUT_CASES = [("sgd", True, 0.3), ("adam", True, 0.5), ("rmsprop", False, 0.2), ("adagrad", True, 0.1)]


@pytest.mark.gpu
@pytest.mark.parametrize("opt,use_bias,p", DROPOUT_CASES)
def test_hip_minibatch_dropout_matches_reference_golden_and_oracle(opt, use_bias, p):
    """cornac_hip_mf_fit_minibatch_dropout: the model against the goldens the real reference produced with dropout = p
    (the masks come from torch's CPU generator on this box exactly as on the box that made the goldens), and the raw
    call against the oracle on ragged batches with duplicate rows and arbitrary masks (k = 5 and 20: a 16-lane group
    covers a row in one and in two passes)"""
    from cornac_amd import MF, _lib

    fx = load_golden("mf_minibatch_dropout")
    m = MF(k=int(fx["k"]), backend="hip-minibatch", optimizer=opt, max_iter=int(fx["epochs"]),
           batch_size=int(fx["batch_size"]), learning_rate=float(fx["lr"]), lambda_reg=float(fx["reg"]),
           use_bias=use_bias, dropout=p, seed=int(fx["seed"])).fit(golden_dataset(fx))
    tag = "%s_p%d%s" % (opt, round(100 * p), "" if use_bias else "_nobias")
    for got, key in ((m.u_factors, "_U"), (m.i_factors, "_V"), (m.u_biases, "_Bu"), (m.i_biases, "_Bi")):
        assert np.abs(np.asarray(got) - fx[tag + key]).max() <= 2e-5, (tag + key, np.abs(np.asarray(got) - fx[tag + key]).max())
    for k in (5, 20):
        rs = np.random.RandomState(k)
        nu, ni, n = 40, 30, 333
        rid, cid = rs.randint(0, nu, n).astype(np.int64), rs.randint(0, ni, n).astype(np.int64)
        val = rs.randint(1, 6, n).astype(np.float32)
        U, V = rs.normal(0, .1, (nu, k)).astype(np.float32), rs.normal(0, .1, (ni, k)).astype(np.float32)
        Bu, Bi = rs.normal(0, .1, nu).astype(np.float32), rs.normal(0, .1, ni).astype(np.float32)
        order = rs.permutation(n)
        batches = [order[s:s + 50] for s in range(0, n, 50)]
        keep_u, keep_i = (rs.rand(n, k) > p).astype(np.uint8), (rs.rand(n, k) > p).astype(np.uint8)
        scale = float(np.float32(1.0) / np.float32(1.0 - p))
        tr = _lib.MfTrainer(rid, cid, val, nu, ni, k)
        tr.set_factors(U, V, Bu, Bi)
        sse = tr.fit_minibatch(order, 50, opt, 0.01, 0.05, 3.0, use_bias, keep_u=keep_u, keep_i=keep_i, keep_scale=scale)
        got = tr.get_factors()
        with pytest.raises(ValueError):
            tr.fit_minibatch(order, 50, opt, 0.01, 0.05, 3.0, use_bias, keep_u=keep_u[:-1], keep_i=keep_i[:-1])
        tr.close()
        keep = [(keep_u[s:s + 50], keep_i[s:s + 50]) for s in range(0, n, 50)]
        want = mf_minibatch_oracle.fit(U, V, Bu, Bi, 3.0, rid, cid, val, batches, opt, 0.01, 0.05, use_bias, keep=keep,
                                       keep_scale=scale)
        for a, b in zip(got[:4] if use_bias else got[:2], want[:4]):
            assert np.abs(a - b).max() <= 2e-5, (opt, k, np.abs(a - b).max())
        assert abs(sse - sum(want[4])) <= 1e-4 * sum(want[4])
