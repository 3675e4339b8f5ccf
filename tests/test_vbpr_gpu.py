"""GPU parity tests of the VBPR minibatch path (cornac_hip_vbpr_*)."""
import numpy as np
import pytest

from cornac_amd import VBPR, _lib
from test_oracle_golden import _vbpr_case

pytestmark = pytest.mark.gpu


def test_vbpr_matches_torch_oracle_and_reference_golden():
    """same batches (mirrored sampler), analytic gradients + dense Adam on the device vs torch autograd +
    torch.optim.Adam: every learned table within 1e-4 of the oracle and of the real reference's golden."""
    from oracle.vbpr_oracle import VBPROracle

    fx, ds, kw = _vbpr_case()
    m = VBPR(verbose=False, **kw).fit(ds)
    o = VBPROracle(**kw).fit(ds)
    for name, key in (("beta_item", "Bi"), ("gamma_user", "Gu"), ("gamma_item", "Gi"), ("theta_user", "Tu"),
                      ("emb_matrix", "E"), ("beta_prime", "Bp"), ("theta_item", "theta_item"),
                      ("visual_bias", "visual_bias")):
        a = np.asarray(getattr(m, name), np.float64).reshape(np.asarray(getattr(o, name)).shape)
        assert np.abs(a - getattr(o, name)).max() <= 1e-4, (name, np.abs(a - getattr(o, name)).max())
        assert np.abs(a - fx[key].reshape(a.shape)).max() <= 1e-4, name
    assert np.abs(m.score(0) - fx["score0"]).max() < 1e-4
    ranked, _ = m.rank(0, k=5)
    assert len(ranked) == ds.num_items and np.all(np.diff(m.score(0)[ranked[:5]]) <= 0)
    assert m.loss_history[-1] < m.loss_history[0]


def test_vbpr_single_step_gradients_against_autograd():
    """one Adam step from random parameters on a batch with duplicate users/items: after step 1 Adam moves
    every parameter with a non-zero gradient by ~lr * sign(grad), so comparing the updated tables checks the
    analytic gradient's support and sign everywhere, and its value through the second step."""
    import torch

    rs = np.random.RandomState(0)
    nu, ni, k, k2, nf, B = 12, 9, 4, 3, 20, 16
    F = rs.uniform(0, 1, (ni, nf)).astype(np.float32)
    P = {"Bi": rs.normal(0, .1, ni), "Gu": rs.normal(0, .3, (nu, k)), "Gi": rs.normal(0, .3, (ni, k)),
         "Tu": rs.normal(0, .3, (nu, k2)), "E": rs.normal(0, .3, (nf, k2)), "Bp": rs.normal(0, .3, nf)}
    P = {n: v.astype(np.float32) for n, v in P.items()}
    u = rs.randint(0, 5, B).astype(np.int32)  # duplicates on purpose
    i = rs.randint(0, ni, B).astype(np.int32)
    j = ((i + 1 + rs.randint(0, ni - 1, B)) % ni).astype(np.int32)
    lr, lw, lb, le = 0.01, 0.02, 0.03, 0.004
    tr = _lib.VbprTrainer(F, nu, ni, k, k2)
    tr.set_params(**P)
    tr.fit_batches(u, i, j, B, lr, lw, lb, le)
    tr.fit_batches(u, i, j, B, lr, lw, lb, le)
    got = tr.get_params()
    tr.close()
    T = {n: torch.tensor(v if n != "Bp" else v.reshape(-1, 1), requires_grad=True) for n, v in P.items()}
    opt = torch.optim.Adam([T[n] for n in ("Bi", "Gu", "Gi", "Tu", "E", "Bp")], lr=lr)
    Ft = torch.tensor(F)
    ul, il, jl = (torch.tensor(x, dtype=torch.long) for x in (u, i, j))
    for _ in range(2):
        gu, tu = T["Gu"][ul], T["Tu"][ul]
        bi_, bj_ = T["Bi"][il], T["Bi"][jl]
        gi, gj = T["Gi"][il], T["Gi"][jl]
        fd = Ft[il] - Ft[jl]
        # [B] + [B, 1] broadcasts to B x B exactly like the reference's Xuij (recom_vbpr.py:242-248)
        X = bi_ - bj_ + (gu * (gi - gj)).sum(1) + (tu * fd.mm(T["E"])).sum(1) + fd.mm(T["Bp"])
        l2 = lambda *ts: sum(t.pow(2).sum() for t in ts) / 2  # noqa: E731
        loss = (-torch.nn.functional.logsigmoid(X).sum() + l2(gu, gi, gj, tu) * lw + l2(bi_) * lb + l2(bj_) * lb / 10
                + l2(T["E"], T["Bp"]) * le)
        opt.zero_grad()
        loss.backward()
        opt.step()
    for n in got:
        want = T[n].detach().numpy().reshape(got[n].shape)
        assert np.abs(got[n] - want).max() < 2e-5, (n, np.abs(got[n] - want).max())


def test_vbpr_errors():
    from cornac_amd.recommender import CornacException
    from conftest import synth_dataset

    with pytest.raises(CornacException):
        VBPR(verbose=False).fit(synth_dataset(20, 15, 100, seed=1))
