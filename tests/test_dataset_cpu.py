"""cornac_amd.Dataset vs the reference's Dataset (cornac/data/dataset.py) — live, draw for draw, where /root/reference is
present — and the reference's own expectations for it (tests/cornac/data/test_dataset.py:33-215) on local data."""
import os

import numpy as np
import pytest

from cornac_amd import Dataset


def _tuples(n_users=25, n_items=18, n=220, seed=3, with_time=False):
    rs = np.random.RandomState(seed)
    keys = rs.permutation(n_users * n_items)[:n]
    out = []
    for k in keys:
        t = ("u%d" % (k // n_items), "i%d" % (k % n_items), float(rs.randint(1, 6)))
        out.append(t + (int(rs.randint(0, 50)),) if with_time else t)   # few distinct timestamps: ties inside a user
    return out


def test_expectations_of_the_reference_dataset_tests(tmp_path):
    data = _tuples()
    ds = Dataset.from_uir(data, seed=123)
    assert ds.matrix.shape == (ds.num_users, ds.num_items) and ds.csr_matrix.has_sorted_indices
    assert ds.uid_map[data[0][0]] == 0 and ds.iid_map[data[0][1]] == 0
    assert set(ds.user_ids) == {t[0] for t in data} and set(ds.item_ids) == {t[1] for t in data}
    assert ds.csr_matrix[0, 0] == data[0][2] == ds.csc_matrix[0, 0] == ds.dok_matrix[0, 0]
    assert len(ds.uir_tuple) == 3 and ds.num_batches(50) == 5
    assert ds.num_user_batches(10) == 3 and ds.num_item_batches(10) == 2
    # iterators: default order is the data order, shuffle changes it
    assert np.array_equal(np.concatenate(list(ds.idx_iter(10, 1))), np.arange(10))
    assert not np.array_equal(np.concatenate(list(ds.idx_iter(100, 1, shuffle=True))), np.arange(100))
    u, i, r = (np.concatenate(x) for x in zip(*ds.uir_iter(batch_size=7)))
    assert np.array_equal(u, ds.uir_tuple[0]) and np.array_equal(i, ds.uir_tuple[1]) and np.array_equal(r, ds.uir_tuple[2])
    assert all((b == 1).all() for _, _, b in ds.uir_iter(batch_size=16, binary=True))
    for bu, bi, br in ds.uir_iter(batch_size=5, num_zeros=2):
        n = len(bu) // 3
        assert (br[:n] > 0).all() and (br[n:] == 0).all() and np.array_equal(bu[n:], bu[:n].repeat(2))
        assert all(ds.dok.get((int(a), int(b)), 0.0) == 0 for a, b in zip(bu[n:], bi[n:]))
    for bu, bp, bn in ds.uij_iter(batch_size=9, neg_sampling="popularity"):
        assert all(ds.dok.get((int(a), int(c)), 0.0) < ds.dok[(int(a), int(b))] for a, b, c in zip(bu, bp, bn))
    with pytest.raises(ValueError):
        next(ds.uij_iter(neg_sampling="bla"))
    assert sorted(np.concatenate(list(ds.user_iter(4))).tolist()) == list(range(ds.num_users))
    assert sorted(np.concatenate(list(ds.item_iter(4, shuffle=True))).tolist()) == list(range(ds.num_items))
    # per-user / per-item views
    assert ds.user_data[0][0][0] == 0 and ds.user_data[0][1][0] == data[0][2]
    assert sum(len(v[0]) for v in ds.item_data.values()) == ds.num_ratings
    with pytest.raises(ValueError):
        ds.chrono_user_data
    dt = Dataset.from_uirt(_tuples(with_time=True))
    assert len(dt.timestamps) == dt.num_ratings
    for items, ratings, times in dt.chrono_user_data.values():
        assert times == sorted(times) and len(items) == len(ratings) == len(times)
    with pytest.raises(ValueError):      # everything is unknown without global id maps
        Dataset.build(data, exclude_unknowns=True)
    # pickle round trip
    path = os.path.join(str(tmp_path), "sub", "ds.pkl")
    ds.save(path)
    back = Dataset.load(path)
    assert back.load_from == path and all(np.array_equal(a, b) for a, b in zip(back.uir_tuple, ds.uir_tuple))


def test_dataset_matches_the_reference_dataset_draw_for_draw():
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    RefDataset = ref_loader.load().Dataset
    for with_time in (False, True):
        data = _tuples(with_time=with_time, seed=8)
        make = (lambda cls: cls.from_uirt(data, seed=5)) if with_time else (lambda cls: cls.from_uir(data, seed=5))
        ref, mine = make(RefDataset), make(Dataset)
        for a, b in zip(ref.uir_tuple, mine.uir_tuple):
            assert np.array_equal(a, b) and a.dtype == b.dtype
        assert list(ref.uid_map.items()) == list(mine.uid_map.items())
        assert (ref.min_rating, ref.max_rating, ref.global_mean) == (mine.min_rating, mine.max_rating, mine.global_mean)
        assert (ref.csr_matrix != mine.csr_matrix).nnz == 0 and (ref.csc_matrix != mine.csc_matrix).nnz == 0
        assert dict(ref.dok_matrix.items()) == dict(mine.dok_matrix.items())

        def same(x, y):
            x, y = list(x), list(y)
            assert len(x) == len(y)
            for bx, by in zip(x, y):
                bx, by = (bx, by) if isinstance(bx, tuple) else ((bx,), (by,))
                for p, q in zip(bx, by):
                    assert np.array_equal(p, q) and np.asarray(p).dtype == np.asarray(q).dtype

        for kw in (dict(batch_size=16), dict(batch_size=7, shuffle=True), dict(batch_size=9, shuffle=True, binary=True),
                   dict(batch_size=11, shuffle=True, num_zeros=3), dict(batch_size=5, num_zeros=1, binary=True)):
            same(ref.reset().uir_iter(**kw), mine.reset().uir_iter(**kw))
        for kw in (dict(batch_size=13, shuffle=True), dict(batch_size=6, neg_sampling="popularity"),
                   dict(batch_size=50, shuffle=True, neg_sampling="popularity")):
            same(ref.reset().uij_iter(**kw), mine.reset().uij_iter(**kw))
        for kw in (dict(batch_size=4), dict(batch_size=3, shuffle=True)):
            same(ref.reset().user_iter(**kw), mine.reset().user_iter(**kw))
            same(ref.reset().item_iter(**kw), mine.reset().item_iter(**kw))
        same(ref.reset().idx_iter(37, 5, True), mine.reset().idx_iter(37, 5, True))
        for name in ("user_data", "item_data") + (("chrono_user_data", "chrono_item_data") if with_time else ()):
            r, m = getattr(ref, name), getattr(mine, name)
            assert [int(k) for k in r.keys()] == list(m.keys())
            for k in m:
                assert tuple(list(map(float, col)) for col in r[k]) == tuple(list(map(float, col)) for col in m[k]), (name, k)
        if with_time:
            assert np.array_equal(ref.timestamps, mine.timestamps)


def test_image_modality_build_places_rows_by_the_global_item_map():
    from collections import OrderedDict

    from cornac_amd import ImageModality, RatioSplit

    rs = np.random.RandomState(4)
    ids = ["i%d" % j for j in rs.permutation(18)]
    F = rs.uniform(-2, 5, (18, 6)).astype(np.float32)
    id_map = OrderedDict(("i%d" % j, n) for n, j in enumerate(rs.permutation(18)[:12]))     # 12 of the 18 items are known
    mod = ImageModality(features=F.copy(), ids=list(ids), normalized=True).build(id_map=id_map)
    lo, hi = F.min(), F.max()
    for raw, new in id_map.items():
        want = (F[ids.index(raw)] - lo) / ((hi - lo) + 1e-10)
        assert np.allclose(mod.features[new], want, rtol=1e-6) and mod.ids[new] == raw
    assert mod.feature_dim == 6 and mod.batch_feature([0, 3]).shape == (2, 6)
    # through the evaluation method: every dataset of the split carries the built modality
    data = _tuples()
    item_ids = sorted({t[1] for t in data})
    feats = rs.normal(size=(len(item_ids), 4)).astype(np.float32)
    split = RatioSplit(data, test_size=0.2, seed=1, item_image=ImageModality(features=feats.copy(), ids=item_ids))
    assert split.test_set.item_image is split.train_set.item_image
    for raw, idx in split.train_set.iid_map.items():
        assert np.array_equal(split.train_set.item_image.features[idx], feats[item_ids.index(raw)])


def test_image_modality_matches_the_reference_modality():
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ref_loader.load()
    import importlib
    from collections import OrderedDict

    from cornac_amd import ImageModality

    RefImage = importlib.import_module("cornac.data").ImageModality
    rs = np.random.RandomState(7)
    for normalized, n_known in ((True, 9), (False, 14), (True, 14)):
        ids = ["x%d" % j for j in rs.permutation(14)]
        F = rs.uniform(-1, 3, (14, 5)).astype(np.float32)
        id_map = OrderedDict(("x%d" % j, n) for n, j in enumerate(rs.permutation(14)[:n_known]))
        ref = RefImage(features=F.copy(), ids=list(ids), normalized=normalized).build(id_map=id_map)
        mine = ImageModality(features=F.copy(), ids=list(ids), normalized=normalized).build(id_map=id_map)
        assert np.array_equal(ref.features, mine.features) and ref.features.dtype == mine.features.dtype
        assert list(ref.ids) == list(mine.ids)


@pytest.mark.parametrize("exclude_unknowns,with_time", [(False, False), (True, False), (False, True), (True, True)])
def test_column_wise_build_equals_the_record_loop(monkeypatch, exclude_unknowns, with_time):
    """large inputs are built column-wise; same arrays, same id maps (content and order), same duplicate count as the
    record-by-record loop, with global maps, unknown ids, duplicate pairs and mixed id types in the data"""
    import warnings
    from collections import OrderedDict

    rs = np.random.RandomState(3)
    n = 6000
    users = [("u%d" % a) if a % 3 else int(a) for a in rs.randint(0, 400, n)]          # strings and ints as raw ids
    items = ["i%d" % b for b in rs.randint(0, 150, n)]
    data = [(u, i, float(r)) + ((int(t),) if with_time else ()) for u, i, r, t in
            zip(users, items, rs.randint(1, 6, n), rs.randint(0, 10 ** 6, n))]

    def run(threshold):
        monkeypatch.setattr(Dataset, "VECTORISED_BUILD_FROM", threshold)
        gu = OrderedDict(("u%d" % a, n) for n, a in enumerate((5, 7, 1, 301, 44)))    # some ids known beforehand
        gi = OrderedDict(("i%d" % b, n) for n, b in enumerate(range(0, 150, 2)))
        if exclude_unknowns:
            gu.update((int(a), len(gu) + n) for n, a in enumerate(range(0, 400, 6)))
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ds = Dataset.build(data, fmt="UIRT" if with_time else "UIR", global_uid_map=gu, global_iid_map=gi, seed=1,
                               exclude_unknowns=exclude_unknowns)
        return ds, [str(x.message) for x in w]

    loop, w_loop = run(10 ** 9)
    fast, w_fast = run(1)
    for a, b in zip(loop.uir_tuple, fast.uir_tuple):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    assert list(loop.uid_map.items()) == list(fast.uid_map.items()) and list(loop.iid_map.items()) == list(fast.iid_map.items())
    assert (loop.num_users, loop.num_items, loop.num_ratings) == (fast.num_users, fast.num_items, fast.num_ratings)
    assert w_loop == w_fast and len(w_loop) == 1 and "duplicated" in w_loop[0]
    if with_time:
        assert np.array_equal(loop.timestamps, fast.timestamps)
    monkeypatch.setattr(Dataset, "VECTORISED_BUILD_FROM", 1)
    with pytest.raises(ValueError):
        Dataset.build(data, exclude_unknowns=True)     # nothing is known without global maps


def test_column_wise_build_equals_the_reference_build():
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    RefDataset = ref_loader.load().Dataset
    rs = np.random.RandomState(9)
    n = 30000
    assert n >= Dataset.VECTORISED_BUILD_FROM
    data = [("u%d" % a, "i%d" % b, float(r)) for a, b, r in zip(rs.randint(0, 3000, n), rs.randint(0, 800, n), rs.randint(1, 6, n))]
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_train, my_train = RefDataset.build(data[:20000]), Dataset.build(data[:20000])
        ref_test = RefDataset.build(data[20000:], global_uid_map=ref_train.uid_map, global_iid_map=ref_train.iid_map,
                                    exclude_unknowns=True)
        my_test = Dataset.build(data[20000:] + data[20000:], global_uid_map=my_train.uid_map, global_iid_map=my_train.iid_map,
                                exclude_unknowns=True)     # doubled: column-wise path, every pair duplicated once
    for ref, mine in ((ref_train, my_train), (ref_test, my_test)):
        for a, b in zip(ref.uir_tuple, mine.uir_tuple):
            assert a.dtype == b.dtype and np.array_equal(a, b)
        assert list(ref.uid_map.items()) == list(mine.uid_map.items()) and list(ref.iid_map.items()) == list(mine.iid_map.items())
        assert (ref.csr_matrix != mine.csr_matrix).nnz == 0
