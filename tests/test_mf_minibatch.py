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
    with pytest.raises(ValueError):
        MF(backend="hip-minibatch", dropout=0.1).fit(golden_dataset(load_golden("tiny")))
    with pytest.raises(KeyError):
        MF(backend="hip-minibatch", optimizer="lbfgs").fit(golden_dataset(load_golden("tiny")))
