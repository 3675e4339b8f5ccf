"""Validates the oracle against the LIVE reference (compiled from /root/reference into
oracle/_ref by oracle/build_ref.py).  Only runs where the reference exists (the build container);
skipped on the GPU box, which relies on the committed golden vectors instead."""
import numpy as np
import pytest


def assert_close(a, b, atol=2e-6, rtol=5e-6):
    """oracle (strict IEEE) vs reference (-ffast-math): a few ulp of the value magnitude"""
    err = np.abs(np.asarray(a) - np.asarray(b)).max()
    assert err <= atol + rtol * np.abs(b).max(), err

from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference not available / oracle/_ref not built")


def _pairs(nu, ni, nnz, seed):
    rs = np.random.RandomState(seed)
    keys = rs.permutation(np.unique(rs.randint(nu, size=3 * nnz).astype(np.int64) * ni + rs.randint(ni, size=3 * nnz)))[:nnz]
    return [(int(k // ni), int(k % ni), float(rs.randint(1, 6))) for k in keys]


@pytest.mark.parametrize("k,use_bias,seed", [(5, True, 1), (16, False, 2), (33, True, 3)])
def test_seeded_models_match_live_reference(oracle, k, use_bias, seed):
    ns = ref_loader.load()
    data = _pairs(80, 60, 1500, seed)
    ds = ns.Dataset.from_uir(data, seed=1)
    kw = dict(k=k, max_iter=6, learning_rate=0.03, lambda_reg=0.01, use_bias=use_bias, seed=seed)
    for ref_cls, or_cls in ((ns.BPR, oracle.BPROracle), (ns.WBPR, oracle.WBPROracle)):
        m, o = ref_cls(**kw).fit(ds), or_cls(**kw).fit(ds)
        assert_close(m.u_factors, o.u_factors)
        assert_close(m.i_factors, o.i_factors)
        assert_close(m.i_biases, o.i_biases)
    m, o = ns.MF(**kw).fit(ds), oracle.MFOracle(**kw).fit(ds)
    assert_close(m.u_factors, o.u_factors)
    assert_close(m.i_factors, o.i_factors)
    assert_close(m.i_biases, o.i_biases)
    assert np.abs(m.score(3) - o.score(3)).max() < 5e-6


def test_host_mirror_matches_reference_dataset_and_rank(oracle):
    """cornac_amd.Dataset builds the same arrays as cornac.data.Dataset, and the oracle's pinned
    rank() agrees with Recommender.rank on a tie-free score vector."""
    from cornac_amd import Dataset

    ns = ref_loader.load()
    data = _pairs(50, 40, 700, 9)
    a, b = ns.Dataset.from_uir(data, seed=1), Dataset.from_uir(data, seed=1)
    for x, y in zip(a.uir_tuple, b.uir_tuple):
        assert x.dtype == y.dtype and np.array_equal(x, y)
    assert np.array_equal(a.matrix.indptr, b.matrix.indptr) and np.array_equal(a.matrix.indices, b.matrix.indices)
    assert a.matrix.indices.dtype == b.matrix.indices.dtype == np.int32
    assert (a.num_users, a.num_items, a.global_mean) == (b.num_users, b.num_items, b.global_mean)
    m = ns.BPR(k=8, max_iter=5, seed=3).fit(a)
    s = m.score(7)
    cand = np.arange(5, 35)
    for k in (-1, 6):
        ref_rank, ref_scores = m.rank(7, item_indices=cand, k=k)
        ranked, scores = oracle.rank(s, a.num_items, len(a.iid_map), item_indices=cand, k=k)
        assert np.array_equal(scores, ref_scores)
        n = len(cand) if k == -1 else k
        assert np.array_equal(ranked[:n], ref_rank[:n])
