"""CPU checks of the WMF oracle (oracle/wmf_oracle.py).  TensorFlow is absent, so the pin is the reference's own WMF
code run over oracle/tf1_shim (tests/golden/wmf_ref.npz here, the live run in tests/test_oracle_vs_reference.py); besides
that: the restated gradients are the gradients of the reference's loss expression (torch autograd of
cornac/models/wmf/wmf.py:44-48), the optimiser follows TF1 Adam's published update, and a regression fixture."""
import numpy as np
import scipy.sparse as sp

from conftest import load_golden
from oracle.wmf_oracle import WmfOracle


def _case(seed=0, nu=60, ni=45, k=7, nnz=500):
    rs = np.random.RandomState(seed)
    keys = rs.permutation(nu * ni)[:nnz]
    u, i = keys // ni, keys % ni
    r = rs.randint(1, 6, nnz).astype(np.float32)
    R = sp.csc_matrix((r, (u, i)), shape=(nu, ni))
    U = rs.normal(0, 0.3, (nu, k)).astype(np.float32)
    V = rs.normal(0, 0.3, (ni, k)).astype(np.float32)
    return R, U, V


def test_first_step_matches_autograd_of_the_reference_loss():
    import torch

    R, U, V = _case()
    ids = np.array([3, 17, 5, 40, 8, 21])
    lu, lv, a, b, lr = 0.03, 0.02, 1.0, 0.05, 0.01
    o = WmfOracle(U, V, R, lu, lv, a, b, lr)
    loss = o.step(ids)
    Ut, Vt = torch.tensor(U, dtype=torch.float64, requires_grad=True), torch.tensor(V, dtype=torch.float64, requires_grad=True)
    Rb = torch.tensor(R[:, ids].toarray(), dtype=torch.float64)
    C = torch.where(Rb != 0, torch.tensor(a, dtype=torch.float64), torch.tensor(b, dtype=torch.float64))
    Vb = Vt[ids]
    L = (C * (Rb - Ut @ Vb.T) ** 2).sum() + lu * 0.5 * (Ut ** 2).sum() + lv * 0.5 * (Vb ** 2).sum()
    L.backward()
    assert abs(loss - L.item()) <= 1e-5 * abs(L.item())
    # step 1 of Adam: m_hat = g, v_hat = g^2  ->  delta = -lr g / (|g| + eps*)   (eps folded, TF1 form)
    for new, old, g in ((o.U, U, Ut.grad.numpy()), (o.V, V, Vt.grad.numpy())):
        g = np.clip(g, -5, 5)
        lr_t = lr * np.sqrt(1 - 0.999) / (1 - 0.9)
        want = old - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
        assert np.abs(new - want).max() < 2e-6
    # rows of V outside the batch have zero gradient and zero moments: they must not move on step 1
    rest = np.setdiff1d(np.arange(V.shape[0]), ids)
    assert np.array_equal(o.V[rest], V[rest])


def test_sparse_adam_moves_all_rows_after_a_row_was_touched():
    """TF1's IndexedSlices Adam: a row touched in step 1 keeps moving in step 2 even if absent from that batch"""
    R, U, V = _case(1)
    o = WmfOracle(U, V, R, lr=0.01)
    o.step(np.array([0, 1, 2]))
    v1 = o.V.copy()
    o.step(np.array([3, 4]))
    assert np.abs(o.V[[0, 1, 2]] - v1[[0, 1, 2]]).max() > 1e-4
    assert np.array_equal(o.V[10:], V[10:])


def test_fixture_regression():
    fx = load_golden("wmf_small")
    R = sp.csc_matrix((fx["ratings"], (fx["users"], fx["items"])), shape=(int(fx["n_users"]), int(fx["n_items"])))
    o = WmfOracle(fx["U0"], fx["V0"], R, float(fx["lambda_u"]), float(fx["lambda_v"]), float(fx["a"]), float(fx["b"]),
                  float(fx["lr"]))
    ptr = fx["batch_ptr"]
    losses = o.fit_batches([fx["batch_ids"][ptr[t]:ptr[t + 1]] for t in range(len(ptr) - 1)])
    assert np.abs(o.U - fx["U"]).max() <= 2e-6 and np.abs(o.V - fx["V"]).max() <= 2e-6
    assert np.allclose(losses, fx["losses"], rtol=1e-6)


def _wmf_ref_case():
    from cornac_amd import Dataset

    fx = load_golden("wmf_ref")
    ds = Dataset.from_uir([(int(u), int(i), float(r)) for u, i, r in zip(fx["users"], fx["items"], fx["ratings"])], seed=123)
    kw = {n: (int(fx[n]) if n in ("k", "max_iter", "batch_size", "seed") else float(fx[n]))
          for n in ("k", "max_iter", "batch_size", "learning_rate", "lambda_u", "lambda_v", "a", "b", "seed")}
    return fx, ds, kw


def test_host_class_and_oracle_reproduce_the_reference_codes_fixture(monkeypatch):
    """tests/golden/wmf_ref.npz: what the reference's OWN WMF code learned (run over oracle/tf1_shim, see
    make_wmf_ref_golden.py) from its own xavier initialisation and item_iter shuffling.  cornac_amd.WMF with the oracle
    as device layer reproduces it to float32 rounding: initialisation, batch order, gradients, Adam, score()."""
    import fake_device

    from cornac_amd import WMF

    fake_device.install(monkeypatch)
    fx, ds, kw = _wmf_ref_case()
    m = WMF(verbose=False, **kw).fit(ds)
    assert np.abs(m.U - fx["U"]).max() < 5e-6 and np.abs(m.V - fx["V"]).max() < 5e-6
    for t, u in enumerate(fx["score_users"]):
        assert np.abs(m.score(int(u)) - fx["scores"][t]).max() < 1e-5
