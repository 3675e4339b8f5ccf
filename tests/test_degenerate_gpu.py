"""Degenerate shapes on the DEVICE kernels (the reference handles them, cornac/models/bpr/recom_bpr.pyx:231-245: a
draw whose negative the user already has is skipped, whatever the catalogue size): one item, one interaction, one to
four users, a user who owns every item (every draw skipped), k = 1; BPR / WBPR / VEBPR / MF, deterministic mode against
the oracle and hogwild mode (the fused kernels' non-owned fallback: fewer interactions than one tile per wave) for
termination and finiteness.  Every case runs under a timeout: a sampler loop that mis-handles a one-element range
would hang."""
import numpy as np
import pytest

from cornac_amd import BPR, MF, VEBPR, WBPR, Dataset, PurchaseViewDataset

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]

SHAPES = {
    "one_user_one_item": [(0, 0)],
    "three_users_one_item": [(0, 0), (1, 0), (2, 0)],
    "one_user_owns_everything": [(0, i) for i in range(5)],
    "single_interaction": [(2, 3)],
    "two_users_three_items": [(0, 0), (0, 1), (1, 2)],
}


def _dataset(name):
    pairs = SHAPES[name]
    rows = [(int(u), int(i), float(1 + (u + 2 * i) % 5)) for u, i in pairs]
    return Dataset.from_uir(rows, seed=1)


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("k", [1, 3])
def test_bpr_wbpr_deterministic_match_the_oracle(oracle, name, k):
    ds = _dataset(name)
    kw = dict(k=k, max_iter=6, learning_rate=0.05, lambda_reg=0.01, seed=5)
    for cls, ocls in ((BPR, oracle.BPROracle), (WBPR, oracle.WBPROracle)):
        m, o = cls(**kw).fit(ds), ocls(**kw).fit(ds)
        assert m.effective_mode == "deterministic"
        assert m.fit_stats[0] == (sum(o.correct), sum(o.skipped)), (cls.__name__, m.fit_stats, o.correct, o.skipped)
        for a, b in ((m.u_factors, o.u_factors), (m.i_factors, o.i_factors), (m.i_biases, o.i_biases)):
            assert np.abs(a - b).max() <= 1e-6
        s = m.score(0)
        assert len(s) == ds.num_items and np.isfinite(s).all()
        ranked, _ = m.rank(0)
        assert sorted(ranked.tolist()) == list(range(ds.num_items))


@pytest.mark.parametrize("name", list(SHAPES))
def test_bpr_hogwild_terminates_and_stays_finite(name):
    ds = _dataset(name)
    nnz = ds.matrix.nnz
    for k in (1, 8, 64):   # 64: the row-wise kernel's non-owned fallback (nnz << one tile per wave)
        m = BPR(k=k, max_iter=4, learning_rate=0.05, lambda_reg=0.01, mode="hogwild", seed=2).fit(ds)
        c, s = m.fit_stats[0]
        assert 0 <= s <= 4 * nnz and 0 <= c <= 4 * nnz - s
        assert np.isfinite(m.u_factors).all() and np.isfinite(m.i_factors).all() and np.isfinite(m.i_biases).all()
        if name in ("one_user_one_item", "three_users_one_item", "one_user_owns_everything"):
            assert s == 4 * nnz, "every negative is one of the user's positives: all draws skipped, nothing moves"


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("k", [1, 5])
def test_mf_both_modes(oracle, name, k):
    ds = _dataset(name)
    kw = dict(k=k, max_iter=5, learning_rate=0.02, lambda_reg=0.02, seed=4)
    m, o = MF(**kw).fit(ds), oracle.MFOracle(**kw).fit(ds)
    for a, b in ((m.u_factors, o.u_factors), (m.i_factors, o.i_factors), (m.u_biases, o.u_biases), (m.i_biases, o.i_biases)):
        assert np.abs(a - b).max() <= 1e-6
    h = MF(mode="hogwild", **kw).fit(ds)
    assert np.isfinite(h.u_factors).all() and np.isfinite(h.i_factors).all()
    assert np.abs(h.u_factors - o.u_factors).max() < 0.5  # same problem, a handful of updates
    assert np.isfinite(h.rate_batch([0], [0])).all()


@pytest.mark.parametrize("views", [[(0, 0)], [(0, 1), (1, 0)], []])
def test_vebpr_tiny(oracle, views):
    pur = [(0, 0, 1.0), (1, 1, 1.0), (1, 2, 1.0)]
    if not views:
        views = [(0, 2)]
    ds = PurchaseViewDataset.build(pur, [(u, i, 1.0) for u, i in views], seed=1)
    kw = dict(k=2, max_iter=5, learning_rate=0.05, lambda_reg=0.01, alpha=0.5, seed=9)
    m, o = VEBPR(**kw).fit(ds), oracle.VEBPROracle(**kw).fit(ds)
    assert m.fit_stats[0] == (sum(o.correct), sum(o.skipped))
    assert np.abs(m.u_factor - o.u_factor).max() <= 1e-6 and np.abs(m.i_factor - o.i_factor).max() <= 1e-6
    h = VEBPR(mode="hogwild", **kw).fit(ds)
    assert np.isfinite(h.u_factor).all() and np.isfinite(h.i_factor).all()
