"""Synthetic interaction sets with the shapes of the reference's benchmark datasets
(SURVEY.md §8d — no network, so MovieLens/Netflix themselves are unavailable).

All generators are pure functions of their seed (np.random.RandomState) so that bench.py, the
tests and the CPU baseline see byte-identical CSR/COO arrays.
"""
import numpy as np

CONFIGS = {
    # name: (n_users, n_items, nnz, zipf_exponent, seed)
    "ml100k": (943, 1682, 80_000, 0.8, 1),            # C1 plumbing (examples/first_example.py)
    "ml20m": (138_493, 26_744, 20_000_263, 0.55, 42),  # C2 headline
    # C3.  Zipf exponent 0.45: the most-rated title then holds ~0.25 % of the ratings, as in the real Netflix Prize
    # set (232 944 of 100 480 507 = 0.23 %); 0.8 would put 2.8 % on one item row
    "netflix": (480_189, 17_770, 100_480_507, 0.45, 43),
}


def zipf_interactions(n_users, n_items, nnz, zipf_a, seed, user_sigma=1.0):
    """Unique (user, item) pairs: item ~ Zipf(zipf_a) over a random item permutation, user activity
    log-normal(sigma).  Returns (users int64, items int64) sorted by (user, item) — the order a
    MovieLens-style file sorted by user gives, which is also CSR order."""
    rs = np.random.RandomState(seed)
    p_item = 1.0 / np.arange(1, n_items + 1) ** zipf_a
    p_item /= p_item.sum()
    perm = rs.permutation(n_items)
    act = rs.lognormal(0.0, user_sigma, n_users)
    p_user = act / act.sum()
    cdf_i = np.cumsum(p_item)
    cdf_u = np.cumsum(p_user)
    keys = np.empty(0, np.int64)
    want = nnz
    while len(keys) < nnz:
        m = int((want - len(keys)) * 1.25) + 1024
        u = np.searchsorted(cdf_u, rs.random_sample(m)).clip(0, n_users - 1)
        i = perm[np.searchsorted(cdf_i, rs.random_sample(m)).clip(0, n_items - 1)]
        keys = np.unique(np.concatenate([keys, u.astype(np.int64) * n_items + i]))
    if len(keys) > nnz:
        keep = rs.choice(len(keys), nnz, replace=False)
        keys = np.sort(keys[keep])
    users, items = keys // n_items, keys % n_items
    return users, items


def make(name, scale=1.0):
    n_users, n_items, nnz, a, seed = CONFIGS[name]
    if scale != 1.0:
        n_users, n_items, nnz = max(8, int(n_users * scale)), max(8, int(n_items * scale)), max(64, int(nnz * scale * scale * 4))
        nnz = min(nnz, n_users * n_items // 4)
    users, items = zipf_interactions(n_users, n_items, nnz, a, seed)
    ratings = np.random.RandomState(seed + 1000).randint(1, 6, size=len(users)).astype(np.float64)
    return n_users, n_items, users, items, ratings


def csr_from_sorted(users, items, n_users):
    """(indptr int32, indices int32) for pairs sorted by (user, item)."""
    counts = np.bincount(users, minlength=n_users)
    indptr = np.zeros(n_users + 1, np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr.astype(np.int32), items.astype(np.int32)
