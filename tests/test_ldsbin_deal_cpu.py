"""The per-epoch item deal of the LDS-bin BPR form (csrc/bpr_ldsbin.inc ldsbin_deal_rank / ldsbin_level_kernel), checked on
its CPU restatement (oracle/cornac_oracle.c oracle_ldsbin_deal; the GPU suite checks device == restatement):

* every (positive, negative) pair of items can meet: the reference draws j uniformly over ALL items for every positive
  (cornac/models/bpr/recom_bpr.pyx:235-238); the binned form draws j inside the positive's bin, so every pair of items
  has to share a bin with probability ~1 / B per epoch — counted here per pair class over 2 000 epoch keys;
* the deal is a partition with one item of every group per bin, and the hot runs level the bins.
"""
import numpy as np
import pytest

from oracle import oracle as orc

B = 256
N_ITEMS = 26744  # the ML-20M shape: 105 groups of 256, 6 strata of 17-18 groups
N_KEYS = 2000


def _zipf_tables(n_items, nnz, expo, seed=0):
    rs = np.random.RandomState(seed)
    p = np.arange(1, n_items + 1, dtype=np.float64) ** -expo
    deg = rs.multinomial(nnz, p / p.sum())
    deg = -np.sort(-deg)  # rank r = item r
    cptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    return np.arange(n_items, dtype=np.int32), cptr, deg


def _bins_over_keys(n_keys, strata_groups, rank_item, cptr, n_hot, n_hot_inter, n_items=N_ITEMS, bins=B):
    n_strata = orc.ldsbin_n_strata(n_items, bins, strata_groups)
    out = np.empty((n_keys, n_items), np.int16)
    rs = np.random.RandomState(123)
    keys = rs.randint(0, 2 ** 32, size=n_keys, dtype=np.uint64)
    for t, key in enumerate(keys):
        out[t] = orc.ldsbin_deal_key(int(key), bins, n_items, n_hot, n_strata, 32, rank_item, cptr, n_hot_inter)[0]
    return out


def _pair_classes(rs, n_pairs, n_items=N_ITEMS, bins=B, strata_groups=16):
    n_groups = (n_items + bins - 1) // bins
    n_strata = orc.ldsbin_n_strata(n_items, bins, strata_groups)
    g_lo = np.array([(s * n_groups) // n_strata for s in range(n_strata + 1)])
    stratum_of_rank = np.searchsorted(g_lo, np.arange(n_items) // bins, side="right") - 1
    cls = {}
    # the round-3 defect: two ranks of one static group of `bins` consecutive ranks
    a = rs.randint(0, n_items - bins, size=n_pairs)
    a -= a % bins
    x = a + rs.randint(0, bins, size=n_pairs)
    y = a + rs.randint(0, bins, size=n_pairs)
    keep = x != y
    cls["same static group"] = (x[keep], y[keep])
    x = rs.randint(0, n_items - 1, size=n_pairs)
    cls["adjacent ranks"] = (x, x + 1)
    x = rs.randint(0, 2 * bins, size=n_pairs)  # the head: the most popular items among themselves
    y = rs.randint(0, 2 * bins, size=n_pairs)
    keep = x != y
    cls["both in the top 2 B"] = (x[keep], y[keep])
    x = rs.randint(0, n_items, size=4 * n_pairs)
    y = rs.randint(0, n_items, size=4 * n_pairs)
    same = (stratum_of_rank[x] == stratum_of_rank[y]) & (x != y)
    cls["same stratum"] = (x[same][:n_pairs], y[same][:n_pairs])
    far = stratum_of_rank[x] != stratum_of_rank[y]
    cls["different strata"] = (x[far][:n_pairs], y[far][:n_pairs])
    return cls


@pytest.fixture(scope="module")
def deals():
    rank_item, cptr, deg = _zipf_tables(N_ITEMS, 20_000_000, 0.55)
    share = 20_000_000 / B
    n_hot = int(np.sum(deg * 1000.0 > share * 100))
    return rank_item, cptr, deg, n_hot, int(deg[:n_hot].sum())


def test_every_pair_class_shares_a_bin_at_the_uniform_rate(deals):
    rank_item, cptr, deg, n_hot, n_hot_inter = deals
    bins = _bins_over_keys(N_KEYS, 16, rank_item, cptr, n_hot, n_hot_inter)
    rs = np.random.RandomState(5)
    for name, (x, y) in _pair_classes(rs, 6000).items():
        assert len(x) >= 2000, name
        together = (bins[:, x] == bins[:, y]).sum(axis=0)  # per pair, over the keys
        rate = together.mean() / N_KEYS
        assert abs(rate * B - 1.0) <= 0.10, "%s: co-bin rate %.5f vs 1/B = %.5f" % (name, rate, 1.0 / B)
        # per pair: ~Poisson(N_KEYS / B = 7.8): no systematically excluded pairs (P(0) = 4e-4 -> a handful at most),
        # no over-dispersion (a pair class that meets in bursts)
        assert (together == 0).mean() <= 5e-3, "%s: %d of %d pairs never met" % (name, (together == 0).sum(), len(x))
        assert 0.75 <= together.var() / together.mean() <= 1.35, name


def test_passing_bins_keep_the_uniform_pair_rate():
    """the same deal with MANY bins (the passing-bin regime of large item tables, csrc/bpr.hip ldsbin_plan: thousands of bins
    of ~100 rows, no hot items in a flat popularity): every pair class still shares a bin at the uniform rate 1 / bins"""
    n_items, bins, n_keys = 50_000, 1024, 1500
    rank_item, cptr, deg = _zipf_tables(n_items, 400_000, 0.2, seed=2)
    where = _bins_over_keys(n_keys, 16, rank_item, cptr, 0, 0, n_items=n_items, bins=bins)
    rs = np.random.RandomState(6)
    for name, (x, y) in _pair_classes(rs, 6000, n_items=n_items, bins=bins).items():
        assert len(x) >= 2000, name
        together = (where[:, x] == where[:, y]).sum(axis=0)
        rate = together.mean() / n_keys
        assert abs(rate * bins - 1.0) <= 0.10, "%s: co-bin rate %.6f vs 1/bins = %.6f" % (name, rate, 1.0 / bins)
        assert 0.75 <= together.var() / together.mean() <= 1.35, name


def test_static_groups_would_exclude_their_own_pairs(deals):
    """strata_groups = 1 permutes inside single groups only, i.e. the round-3 deal: the pair-coverage test above must
    be able to see that defect."""
    rank_item, cptr, deg, n_hot, n_hot_inter = deals
    bins = _bins_over_keys(200, 1, rank_item, cptr, n_hot, n_hot_inter)
    x, y = _pair_classes(np.random.RandomState(5), 3000, strata_groups=1)["same static group"]
    assert (bins[:, x] == bins[:, y]).sum() == 0


@pytest.mark.parametrize("n_items,bins,groups", [(N_ITEMS, B, 16), (3003, 256, 16), (3003, 256, 4), (1000, 512, 16),
                                                 (70_000, 1024, 8), (257, 256, 16), (256, 256, 2), (300_000, 2816, 16)])
def test_deal_is_a_partition_with_one_item_per_group_and_levelled_bins(n_items, bins, groups):
    rank_item, cptr, deg = _zipf_tables(n_items, 40 * n_items * 8, 0.7, seed=3)
    share = cptr[-1] / bins
    n_hot = int(np.sum(deg * 1000.0 > share * 100))
    H = int(deg[:n_hot].sum())
    n_groups = (n_items + bins - 1) // bins
    n_strata = orc.ldsbin_n_strata(n_items, bins, groups)
    for key in (0, 1, 0xDEADBEEF, 0xFFFFFFFF):
        bin_of, cold, off = orc.ldsbin_deal_key(key, bins, n_items, n_hot, n_strata, 32, rank_item, cptr, H)
        assert bin_of.min() >= 0 and bin_of.max() < bins
        per_bin = np.bincount(bin_of, minlength=bins)
        assert per_bin.max() <= n_groups and per_bin.min() >= n_groups - 1  # one item of every (full) group
        # the cold masses are the degrees of the bin's non-hot items; every hot interaction belongs to exactly one bin
        want = np.bincount(bin_of[n_hot:], weights=deg[n_hot:].astype(np.float64), minlength=bins)
        assert np.array_equal(cold.astype(np.int64), want.astype(np.int64))
        assert off[0] == 0 and off[-1] == H and np.all(np.diff(off.astype(np.int64)) >= 0)
        # levelling: no bin that received hot interactions is (16 cold + 32 hot)-heavier than the level reached by
        # the others by more than one hot draw + the round-robin remainder
        hot = np.diff(off.astype(np.int64))
        t = 16 * cold.astype(np.int64) + 32 * hot
        filled = hot > 1
        if filled.sum() > 1:
            assert t[filled].max() - t[filled].min() <= 32 * 2 + 16, (t[filled].max(), t[filled].min())
            assert np.all(16 * cold.astype(np.int64)[~filled] >= t[filled].min() - 32 * 2 - 16)


def test_even_split_when_hot_draws_are_not_priced(deals):
    rank_item, cptr, deg, n_hot, H = deals
    _, cold, off = orc.ldsbin_deal_key(77, B, N_ITEMS, n_hot, 6, 0, rank_item, cptr, H)
    hot = np.diff(off.astype(np.int64))
    assert hot.max() - hot.min() <= 1 and hot.sum() == H
