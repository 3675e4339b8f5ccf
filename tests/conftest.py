import os
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    warnings.filterwarnings("ignore", message=".*smallest subnormal.*")


def pytest_sessionstart(session):
    """a checkout without build products (they are git-ignored): build the library and the oracle once, as
    `__graft_entry__.build()` does, instead of failing every test that loads them"""
    lib = os.path.join(ROOT, "cornac_amd", "lib", "libcornac_hip.so")
    if not os.path.exists(lib):
        try:
            import __graft_entry__

            __graft_entry__.build()
        except Exception as e:  # the tests that need the library will say what is missing
            warnings.warn("could not build libcornac_hip.so: %s" % e)


def _gpu_count():
    try:
        from cornac_amd import _lib

        return _lib.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_dataset(fx):
    from cornac_amd import Dataset

    triplets = [(int(u), int(i), float(r)) for u, i, r in zip(fx["users"], fx["items"], fx["ratings"])]
    return Dataset.from_uir(triplets, seed=123)


def synth_dataset(n_users, n_items, nnz, zipf=0.8, seed=0, shuffle=True):
    """small synthetic interaction set as a cornac_amd.Dataset (insertion order shuffled)"""
    from cornac_amd import Dataset

    rs = np.random.RandomState(seed)
    p = 1.0 / np.arange(1, n_items + 1) ** zipf
    p /= p.sum()
    keys = np.unique(rs.randint(n_users, size=nnz * 2).astype(np.int64) * n_items + rs.choice(n_items, nnz * 2, p=p))
    keys = rs.permutation(keys)[:nnz] if shuffle else keys[:nnz]
    u, i = keys // n_items, keys % n_items
    # ratings with user/item structure so that MF has something to learn
    bu, bi = rs.normal(0, 0.8, n_users), rs.normal(0, 0.8, n_items)
    r = np.clip(np.rint(3.0 + bu[u] + bi[i] + rs.normal(0, 0.5, len(u))), 1, 5).astype(float)
    return Dataset.from_uir(list(zip(u.tolist(), i.tolist(), r.tolist())), seed=123)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc
