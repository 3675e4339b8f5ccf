"""GPU parity tests of the WMF minibatch path (cornac_hip_wmf_*) against oracle/wmf_oracle.py."""
import numpy as np
import pytest
import scipy.sparse as sp

from cornac_amd import WMF, _lib
from conftest import load_golden, synth_dataset
from oracle.wmf_oracle import WmfOracle

pytestmark = pytest.mark.gpu


def _run_both(R, U, V, batches, lu, lv, a, b, lr):
    o = WmfOracle(U, V, R, lu, lv, a, b, lr)
    lo = np.array(o.fit_batches(batches))
    tr = _lib.WmfTrainer(R, U.shape[1])
    tr.set_factors(U, V)
    lg = tr.fit_batches(batches, lu, lv, a, b, lr)
    Ug, Vg = tr.get_factors()
    tr.close()
    return o, lo, Ug, Vg, lg


@pytest.mark.parametrize("nu,ni,k,bs", [(300, 200, 24, 64), (129, 257, 5, 128), (1000, 90, 200, 37), (64, 3, 33, 2),
                                         (700, 300, 128, 128), (517, 260, 100, 77), (40000, 256, 128, 128),
                                         (128, 140, 128, 128), (129, 140, 120, 100), (5, 130, 128, 128), (33000, 200, 97, 128)])
def test_steps_match_oracle(nu, ni, k, bs):
    """k in 97..128 takes the wave-specialised kernel (wmf_ws.inc): full and ragged user tiles, ragged batches, and at
    40 000 users several tiles per workgroup (the Adam sweep of a tile runs beside the products of the next one); exactly one
    tile, one row over a tile, fewer users than one slice (the tables are padded to whole tiles + one: rows that do not exist
    must stay out of the result and out of the loss), 33 000 users = 258 tiles on 256 workgroups (two of them sweep a real
    previous tile, the others only the scratch tile)"""
    rs = np.random.RandomState(nu + k)
    nnz = min(nu * ni // 3, max(6000, 4 * nu))
    keys = rs.permutation(nu * ni)[:nnz]
    u, i = keys // ni, keys % ni
    R = sp.csc_matrix((rs.randint(1, 6, nnz).astype(np.float32), (u, i)), shape=(nu, ni))
    U = rs.normal(0, 0.2, (nu, k)).astype(np.float32)
    V = rs.normal(0, 0.2, (ni, k)).astype(np.float32)
    batches = []
    for _ in range(3):
        perm = rs.permutation(ni)
        batches += [perm[s:s + bs] for s in range(0, ni, bs)]
    o, lo, Ug, Vg, lg = _run_both(R, U, V, batches, 0.02, 0.03, 1.0, 0.01, 0.005)
    assert np.abs(Ug - o.U).max() <= 1e-4, np.abs(Ug - o.U).max()
    assert np.abs(Vg - o.V).max() <= 1e-4, np.abs(Vg - o.V).max()
    assert np.allclose(lg, lo, rtol=2e-5), np.abs(lg / lo - 1).max()


def test_steps_match_oracle_without_the_unobserved_weight():
    """b = 0: G carries nothing to recover a prediction from, the fix-up recomputes the dot products (both fused kernels)"""
    for nu, ni, k in [(300, 150, 128), (300, 150, 40)]:
        rs = np.random.RandomState(k)
        keys = rs.permutation(nu * ni)[:5000]
        R = sp.csc_matrix((rs.randint(1, 6, 5000).astype(np.float32), (keys // ni, keys % ni)), shape=(nu, ni))
        U = rs.normal(0, 0.2, (nu, k)).astype(np.float32)
        V = rs.normal(0, 0.2, (ni, k)).astype(np.float32)
        perm = rs.permutation(ni)
        batches = [perm[s:s + 128] for s in range(0, ni, 128)] * 2
        o, lo, Ug, Vg, lg = _run_both(R, U, V, batches, 0.02, 0.03, 1.0, 0.0, 0.005)
        assert np.abs(Ug - o.U).max() <= 1e-4 and np.abs(Vg - o.V).max() <= 1e-4
        assert np.allclose(lg, lo, rtol=2e-5)


def test_fixture_and_model_surface():
    fx = load_golden("wmf_small")
    from cornac_amd import Dataset

    ds = Dataset.from_uir([(int(u), int(i), float(r)) for u, i, r in zip(fx["users"], fx["items"], fx["ratings"])], seed=123)
    kw = dict(k=int(fx["k"]), lambda_u=float(fx["lambda_u"]), lambda_v=float(fx["lambda_v"]), a=float(fx["a"]),
              b=float(fx["b"]), learning_rate=float(fx["lr"]), batch_size=int(fx["batch_size"]), max_iter=int(fx["max_iter"]))
    m = WMF(verbose=False, seed=7, init_params={"U": fx["U0"].copy(), "V": fx["V0"].copy()}, **kw).fit(ds)
    assert np.abs(m.U - fx["U"]).max() <= 1e-4 and np.abs(m.V - fx["V"]).max() <= 1e-4
    assert m.loss_history[-1] < m.loss_history[0]
    s = m.score(0)
    assert np.abs(s - m.V @ m.U[0]).max() < 1e-5
    ranked, scores = m.rank(0, k=5)
    assert len(ranked) == len(scores) and scores[ranked[0]] == scores.max() and np.all(np.diff(scores[ranked[:5]]) <= 0)
    # explicit zeros and empty columns: the weights fall back to b and nothing breaks
    R = sp.csc_matrix((np.array([0.0, 2.0], np.float32), (np.array([0, 1]), np.array([0, 0]))), shape=(4, 3))
    U = np.full((4, 2), 0.1, np.float32); V = np.full((3, 2), 0.2, np.float32)
    o, lo, Ug, Vg, lg = _run_both(R, U, V, [np.array([0, 2]), np.array([1])], 0.01, 0.01, 1.0, 0.5, 0.01)
    assert np.abs(Ug - o.U).max() <= 1e-6 and np.abs(Vg - o.V).max() <= 1e-6 and np.allclose(lg, lo, rtol=1e-6)


def test_training_learns_and_errors():
    ds = synth_dataset(400, 150, 6000, seed=3)
    m = WMF(k=16, max_iter=15, learning_rate=0.01, verbose=False, seed=1).fit(ds)
    assert m.loss_history[-1] < 0.85 * m.loss_history[0]
    with pytest.raises(ValueError):
        WMF(k=4, batch_size=500, verbose=False).fit(ds)
    tr = _lib.WmfTrainer(ds.csc_matrix, 4)
    with pytest.raises(_lib.HipError):
        tr.fit_batches([np.array([10 ** 6])], 0.1, 0.1, 1, 0.01, 0.01)
    tr.close()


def test_hip_wmf_matches_the_reference_codes_fixture():
    """cornac_amd.WMF on the device against tests/golden/wmf_ref.npz — U, V and score() as learned by the reference's own
    WMF code (cornac/models/wmf/recom_wmf.py + wmf.py over oracle/tf1_shim; tests/golden/make_wmf_ref_golden.py) from its
    own initialisation and batch order: north_star's 1e-4"""
    from cornac_amd import Dataset

    fx = load_golden("wmf_ref")
    ds = Dataset.from_uir([(int(u), int(i), float(r)) for u, i, r in zip(fx["users"], fx["items"], fx["ratings"])], seed=123)
    kw = {n: (int(fx[n]) if n in ("k", "max_iter", "batch_size", "seed") else float(fx[n]))
          for n in ("k", "max_iter", "batch_size", "learning_rate", "lambda_u", "lambda_v", "a", "b", "seed")}
    m = WMF(verbose=False, **kw).fit(ds)
    assert np.abs(m.U - fx["U"]).max() <= 1e-4 and np.abs(m.V - fx["V"]).max() <= 1e-4
    for t, u in enumerate(fx["score_users"]):
        assert np.abs(m.score(int(u)) - fx["scores"][t]).max() <= 1e-4
