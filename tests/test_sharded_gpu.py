"""GPU tests of the row-sharded item table path (multi-GPU regime 2): the sample / gather / apply /
scatter-add kernels behind cornac_hip_bpr_{sample_triplets,apply_triplets,gather_rows,scatter_add_rows}
and the RowShardedBprTrainer loop on one rank through a real NCCL (RCCL) group of size 1."""
import os

import numpy as np
import pytest

from cornac_amd import _lib
from conftest import synth_dataset

pytestmark = pytest.mark.gpu


def _trainer(ds, k):
    X = ds.matrix
    return _lib.BprTrainer(X.indptr, X.indices, ds.num_users, ds.num_items, len(ds.uid_map), len(ds.iid_map), k)


def test_sample_triplets_match_the_cpu_restatement_of_the_hogwild_sampler(oracle):
    import torch

    ds = synth_dataset(300, 50, 6000, zipf=0.5, seed=8)
    X = ds.matrix
    indptr, indices, user_ids = oracle.csr_arrays(ds)
    tr = _trainer(ds, 8)
    tr.seed_hogwild(0xABCDEF0123)
    n = 2 * X.nnz + 777  # crosses two epoch boundaries of the sample counter
    dev = torch.device("cuda", 0)
    u, i, j = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    torch.cuda.synchronize()
    tr.sample_triplets(n, u.data_ptr(), i.data_ptr(), j.data_ptr())
    c, s = tr.sync()
    u, i, j = u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()
    tr.close()
    want_u, want_i, want_j = [], [], []
    for epoch, cnt in enumerate((X.nnz, X.nnz, 777)):
        ii, jj = oracle.hogwild_sample(0xABCDEF0123, epoch, 0, cnt, X.nnz, ds.num_items)
        uu = user_ids[ii]
        skip = np.asarray(X[uu, jj]).ravel() != 0
        want_u.append(np.where(skip, -1, uu)); want_i.append(np.where(skip, -1, indices[ii])); want_j.append(np.where(skip, -1, jj))
    assert np.array_equal(u, np.concatenate(want_u)) and np.array_equal(i, np.concatenate(want_i))
    assert np.array_equal(j, np.concatenate(want_j))
    assert s == int((u < 0).sum()) and 0 < s < n // 2


@pytest.mark.parametrize("k", [5, 64, 100])
def test_apply_gather_scatter_kernels(k):
    import torch

    rs = np.random.RandomState(k)
    ds = synth_dataset(200, 80, 3000, seed=2)
    nu = len(ds.uid_map)
    tr = _trainer(ds, k)
    U = rs.normal(0, 0.3, (nu, k)).astype(np.float32)
    tr.set_factors(U, np.zeros((len(ds.iid_map), k), np.float32), np.zeros(len(ds.iid_map), np.float32))
    dev = torch.device("cuda", 0)
    n_slots, n = 150, 60
    rows = rs.normal(0, 0.3, (n_slots, k)).astype(np.float32)
    bias = rs.normal(0, 0.3, n_slots).astype(np.float32)
    # conflict-free batch: distinct users, distinct slots -> exact sequential semantics
    users = rs.permutation(nu)[:n].astype(np.int32)
    slots = rs.permutation(n_slots)[: 2 * n].astype(np.int32)
    si, sj = slots[:n].copy(), slots[n:].copy()
    users[5] = -1  # ignored entry
    t_rows = torch.tensor(rows, device=dev)
    stride = 32
    t_bias = torch.zeros(n_slots, stride, device=dev)
    t_bias[:, 0] = torch.tensor(bias, device=dev)
    lr, reg = 0.05, 0.01
    t_u, t_si, t_sj = (torch.tensor(x, device=dev) for x in (users, si, sj))
    torch.cuda.synchronize()  # the handle runs on its own stream here: inputs must be complete before the launch
    tr.apply_triplets(t_u.data_ptr(), t_si.data_ptr(), t_sj.data_ptr(), n, t_rows.data_ptr(), t_bias.data_ptr(), stride,
                      lr, reg, True)
    c, _ = tr.sync()
    U2 = tr.get_factors()[0]
    wantU, wantR, wantB, correct = U.astype(np.float64), rows.astype(np.float64), bias.astype(np.float64), 0
    for t in range(n):
        u = users[t]
        if u < 0:
            continue
        a, b = si[t], sj[t]
        uf, vi, vj = U[u].astype(np.float64), rows[a].astype(np.float64), rows[b].astype(np.float64)
        z = 1.0 / (1.0 + np.exp(bias[a] - bias[b] + uf @ (vi - vj)))
        correct += z < 0.5
        wantU[u] += lr * (z * (vi - vj) - reg * uf)
        wantR[a] += lr * (z * uf - reg * vi)
        wantR[b] += lr * (-z * uf - reg * vj)
        wantB[a] += lr * (z - reg * bias[a])
        wantB[b] += lr * (-z - reg * bias[b])
    assert c == correct
    assert np.abs(U2 - wantU).max() < 2e-6 and np.abs(t_rows.cpu().numpy() - wantR).max() < 2e-6
    assert np.abs(t_bias[:, 0].cpu().numpy() - wantB).max() < 2e-6 and float(t_bias[:, 1:].abs().max()) == 0.0
    # gather / scatter-add with repeated ids
    table = torch.tensor(rs.normal(0, 1, (90, k)).astype(np.float32), device=dev)
    ids = torch.tensor(rs.randint(0, 90, 400).astype(np.int32), device=dev)
    out = torch.empty(400, k, device=dev)
    delta = torch.tensor(rs.normal(0, 1, (400, k)).astype(np.float32), device=dev)
    torch.cuda.synchronize()
    tr.gather_rows(table.data_ptr(), ids.data_ptr(), 400, k, out.data_ptr())
    tr.sync()
    assert torch.equal(out, table[ids.long()])
    want = table.double().index_add(0, ids.long(), delta.double())
    torch.cuda.synchronize()
    tr.scatter_add_rows(table.data_ptr(), ids.data_ptr(), 400, k, delta.data_ptr())
    tr.sync()
    assert (table.double() - want).abs().max() < 1e-5
    tr.close()


def _big_trainer(k, seed=3):
    from cornac_amd import synth

    n_users, n_items = 6000, 3000
    users, items = synth.zipf_interactions(n_users, n_items, 700_000, 0.8, seed)
    indptr, indices = synth.csr_from_sorted(users, items, n_users)
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    return tr, indptr, indices, n_users, n_items


@pytest.mark.parametrize("k", [64, 100])
def test_emit_triplets_are_valid_owned_draws(k):
    """the EMIT launch of the owned kernel: every emitted triplet is (user, one of its positives, a non-positive),
    emitted + skipped = draws, padding slots are -1, and a non-shared user only ever appears in ONE wave's slots"""
    import torch

    tr, indptr, indices, n_users, n_items = _big_trainer(k)
    tr.seed_hogwild(99)
    nnz = len(indices)
    # a launch covers whole 64-sample tiles of every wave, so only whole epochs have an exact draw count: one third of
    # an epoch, then the rest of it plus a whole second epoch (the second call crosses the epoch boundary)
    n1 = nnz // 3
    n2 = 2 * nnz - n1
    n = n1 + n2
    cap = tr.staged_slots(n2)
    assert cap > 0 and cap % 64 == 0 and tr.staged_slots(n1) <= cap
    dev = torch.device("cuda", 0)
    parts, skipped = [], 0
    for nn in (n1, n2):
        u, i, j = (torch.full((cap,), 7, dtype=torch.int32, device=dev) for _ in range(3))
        torch.cuda.synchronize()
        n_slots = tr.emit_triplets(nn, u.data_ptr(), i.data_ptr(), j.data_ptr(), cap)
        assert 0 <= n_slots <= cap
        skipped += tr.sync()[1]
        parts.append([x[:n_slots].cpu().numpy() for x in (u, i, j)])
    u, i, j = (np.concatenate([p[q] for p in parts]) for q in range(3))
    n_slots = len(u)
    ok = i >= 0
    assert np.array_equal(ok, j >= 0) and np.array_equal(ok, u >= 0)
    assert int(ok.sum()) + skipped == n and 0 < skipped < 0.2 * n
    shared = (u & 0x40000000) != 0
    uu = np.where(ok, u & 0x3FFFFFFF, 0)
    assert uu.max() < n_users and i[ok].max() < n_items and j[ok].max() < n_items
    import scipy.sparse as sp

    X = sp.csr_matrix((np.ones(nnz, np.int8), indices, indptr), shape=(n_users, n_items))
    assert np.asarray(X[uu[ok], i[ok]]).all() and not np.asarray(X[uu[ok], j[ok]]).any()
    # slot -> wave: slots are laid out [tile][wave][lane]
    W = tr.staged_slots(1) // (3 * 64)
    wave = (np.arange(n_slots) // 64) % W
    excl = ok & ~shared
    first = np.full(n_users, -1, np.int64)
    first[uu[excl]] = wave[excl]                      # any one wave the user appears in ...
    assert np.array_equal(first[uu[excl]], wave[excl])  # ... is the only one
    deg = np.diff(indptr)
    assert not shared[ok].any() or deg[uu[ok & shared]].min() > deg[uu[excl]].max() // 2
    tr.close()


@pytest.mark.parametrize("k", [40, 64, 128, 160, 200])
def test_apply_staged_equals_the_sequential_update_on_a_conflict_free_batch(k):
    import torch

    rs = np.random.RandomState(k)
    tr, indptr, indices, nu, n_items = _big_trainer(k)
    U = rs.normal(0, 0.3, (nu, k)).astype(np.float32)
    tr.set_factors(U, np.zeros((n_items, k), np.float32), np.zeros(n_items, np.float32))
    dev = torch.device("cuda", 0)
    unit = tr.staged_slots(1) // 3          # waves * 64: one tile of every wave
    assert unit > 0 and unit % 64 == 0
    n_slots_tab, n, total = 150, 60, 2 * unit
    rows = rs.normal(0, 0.3, (n_slots_tab, k)).astype(np.float32)
    bias = rs.normal(0, 0.3, n_slots_tab).astype(np.float32)
    users = rs.permutation(nu)[:n].astype(np.int32)
    slots = rs.permutation(n_slots_tab)[: 2 * n].astype(np.int32)
    si, sj = slots[:n].copy(), slots[n:].copy()
    where = rs.permutation(total)[:n]
    au, ai, aj = (np.full(total, -1, np.int32) for _ in range(3))
    au[where], ai[where], aj[where] = users, si, sj
    au[where[3]] |= 0x40000000            # a "shared" user: same arithmetic through the atomic path
    t_rows = torch.tensor(rows, device=dev)
    stride = 32
    t_bias = torch.zeros(n_slots_tab, stride, device=dev)
    t_bias[:, 0] = torch.tensor(bias, device=dev)
    lr, reg = 0.05, 0.01
    t_u, t_si, t_sj = (torch.tensor(x, device=dev) for x in (au, ai, aj))
    torch.cuda.synchronize()
    # apply_staged follows an emit on the handle (ownership tables of the same grid)
    tr.seed_hogwild(1)
    scratch = torch.empty(3, tr.staged_slots(64), dtype=torch.int32, device=dev)
    tr.emit_triplets(64, scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[2].data_ptr(), scratch.shape[1])
    tr.sync()
    tr.apply_staged(t_u.data_ptr(), t_si.data_ptr(), t_sj.data_ptr(), total, t_rows.data_ptr(), t_bias.data_ptr(), stride,
                    lr, reg, True)
    c, _ = tr.sync()
    U2 = tr.get_factors()[0]
    wantU, wantR, wantB, correct = U.astype(np.float64), rows.astype(np.float64), bias.astype(np.float64), 0
    for t in range(n):
        u, a, b = users[t], si[t], sj[t]
        uf, vi, vj = U[u].astype(np.float64), rows[a].astype(np.float64), rows[b].astype(np.float64)
        z = 1.0 / (1.0 + np.exp(bias[a] - bias[b] + uf @ (vi - vj)))
        correct += z < 0.5
        wantU[u] += lr * (z * (vi - vj) - reg * uf)
        wantR[a] += lr * (z * uf - reg * vi)
        wantR[b] += lr * (-z * uf - reg * vj)
        wantB[a] += lr * (z - reg * bias[a])
        wantB[b] += lr * (-z - reg * bias[b])
    assert c == correct
    assert np.abs(U2 - wantU).max() < 2e-6 and np.abs(t_rows.cpu().numpy() - wantR).max() < 2e-6
    assert np.abs(t_bias[:, 0].cpu().numpy() - wantB).max() < 2e-6 and float(t_bias[:, 1:].abs().max()) == 0.0
    tr.close()


def test_dedupe_and_scatter_diff_kernels():
    import torch

    from cornac_amd.dist import DeviceRowOps, RowShardedItemTable

    rs = np.random.RandomState(4)
    tr, *_ = _big_trainer(64)
    dev = torch.device("cuda", 0)
    tr.set_stream(torch.cuda.current_stream(dev).cuda_stream)   # as RowShardedBprTrainer does: one stream for both
    n_items, k, world = 1001, 8, 1
    m = 5000
    i = rs.randint(0, n_items, m).astype(np.int32)
    j = rs.randint(0, n_items, m).astype(np.int32)
    skip = rs.rand(m) < 0.1
    i[skip] = j[skip] = -1
    for ops in (DeviceRowOps(tr), None):
        class Plain:   # the torch formulation (the gloo tests' path): gather / scatter_add only
            def gather(self, table, ids, out): out.copy_(table[ids.long()])
            def scatter_add(self, table, ids, delta): table.index_add_(0, ids.long(), delta)
        t = RowShardedItemTable(n_items, k, dev, ops or Plain())
        assert not t.collective and t.world == world
        ti, tj = torch.tensor(i, device=dev), torch.tensor(j, device=dev)
        torch.cuda.synchronize()
        local_rows, slot_i, slot_j, sc, rc = t.dedupe_items(ti, tj)
        tr.sync()
        torch.cuda.synchronize()
        uniq = np.unique(np.concatenate([i[~skip], j[~skip]]))
        assert sc == [len(uniq)] == rc and np.array_equal(local_rows.cpu().numpy(), uniq)
        assert np.array_equal(uniq[slot_i.cpu().numpy()[~skip]], i[~skip])
        assert np.array_equal(uniq[slot_j.cpu().numpy()[~skip]], j[~skip])
    # scatter_diff with repeated ids and a per-row scale
    table = torch.tensor(rs.normal(0, 1, (90, k)).astype(np.float32), device=dev)
    ids = torch.tensor(rs.randint(0, 90, 400).astype(np.int32), device=dev)
    now = torch.tensor(rs.normal(0, 1, (400, k)).astype(np.float32), device=dev)
    before = torch.tensor(rs.normal(0, 1, (400, k)).astype(np.float32), device=dev)
    scale = torch.tensor(rs.uniform(0.5, 1, 400).astype(np.float32), device=dev)
    want = table.double().index_add(0, ids.long(), (now.double() - before.double()) * scale.double().unsqueeze(1))
    torch.cuda.synchronize()
    tr.scatter_diff_rows(table.data_ptr(), ids.data_ptr(), 400, k, now.data_ptr(), before.data_ptr(), scale.data_ptr())
    tr.sync()
    assert (table.double() - want).abs().max() < 1e-5
    tr.close()


def test_row_sharded_trainer_single_rank_through_rccl():
    """the whole regime-2 loop on one rank with a real NCCL group (all_to_all_single through RCCL): learns like
    the fused hogwild kernel on the same data, and with reg = 0 conserves the column sums of V / the sum of B
    (every triplet adds +d to row i and -d to row j), which a mis-routed or double-applied delta would break."""
    import torch
    import torch.distributed as dist

    from cornac_amd import synth
    from cornac_amd.dist import RowShardedBprTrainer

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_users, n_items, k = 6000, 3000, 64
    users, items = synth.zipf_interactions(n_users, n_items, 700_000, 0.8, 3)
    indptr, indices = synth.csr_from_sorted(users, items, n_users)
    rs = np.random.RandomState(0)
    U = rs.normal(0, 0.1, (n_users, k)).astype(np.float32)
    V = rs.normal(0, 0.1, (n_items, k)).astype(np.float32)
    epochs, lr = 3, 0.05
    ref = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    ref.set_factors(U, V, np.zeros(n_items, np.float32))
    ref.seed_hogwild(5)
    c_ref, s_ref = ref.fit_epochs(epochs, lr, 0.0, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags=128)
    ref.close()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
        tr.set_factors(U, None, None)
        tr.seed_hogwild(5)
        sh = RowShardedBprTrainer(tr, n_items, k, dev, micro_batch=100_000)
        assert sh.table.collective
        sh.load_items(V, np.zeros(n_items, np.float32))
        sh.run(epochs * len(indices), lr, 0.0)
        c, s = sh.finish()
        V2t, B2t = sh.table.gather_full()
        V2, B2 = V2t.cpu().numpy(), B2t.cpu().numpy()
        fetched, trip = sh.rows_fetched, sh.triplets
        tr.close()
    finally:
        dist.destroy_process_group()
    n = epochs * len(indices)
    assert trip == n - s and 0 < s < 0.2 * n
    assert abs(c / trip - c_ref / (n - s_ref)) < 0.02, (c / trip, c_ref / (n - s_ref))
    assert fetched <= 2 * trip and fetched >= n_items
    moved = np.abs(V2.astype(np.float64) - V).sum(0)
    assert moved.min() > 1.0
    assert np.abs(V2.astype(np.float64).sum(0) - V.astype(np.float64).sum(0)).max() <= 1e-4 * moved.max() + 1e-3
    assert abs(float(B2.astype(np.float64).sum())) <= 1e-4 * np.abs(B2).sum() + 1e-3


@pytest.mark.parametrize("k", [5, 8])
def test_fused_table_delta_kernels_equal_the_torch_algebra(k):
    """ItemTableReplica's two elementwise passes as HIP kernels vs the torch formulation (the gloo tests' path),
    with a hand-made 'all-reduced' bucket in between: summed deltas of 3 virtual ranks and their touch counts.
    k = 5: the scalar form of the kernels; k = 8: the 16-byte form (k % 4 == 0)."""
    import torch

    ds = synth_dataset(50, 40, 300, seed=1)
    tr = _trainer(ds, 4)
    dev = torch.device("cuda", 0)
    n = 37
    g = torch.Generator(device="cpu").manual_seed(0)
    base = torch.randn(n * k + n, generator=g).to(dev)
    flat = base.clone()
    touched_rows = torch.tensor([0, 3, 4, 20, 36])
    flat[: n * k].view(n, k)[touched_rows] += torch.randn(len(touched_rows), k, generator=g).to(dev)
    flat[n * k + 7] += 0.5
    bucket = torch.empty(n * k + 3 * n, device=dev)
    local = torch.empty(n * k + n, device=dev)
    torch.cuda.synchronize()
    tr.table_delta_begin(flat.data_ptr(), base.data_ptr(), n, k, bucket.data_ptr(), local.data_ptr())
    tr.sync()
    d = flat - base
    assert torch.equal(bucket[: n * k + n], d) and torch.equal(local, d)
    assert torch.equal(bucket[n * k + n: n * k + 2 * n], (d[: n * k].view(n, k) != 0).any(1).float())
    assert torch.equal(bucket[n * k + 2 * n:], (d[n * k:] != 0).float())
    # pretend two more ranks contributed: other deltas on overlapping and distinct rows
    other = torch.zeros_like(bucket)
    other[: n * k].view(n, k)[[0, 1, 36]] = torch.randn(3, k, generator=g).to(dev)
    other[n * k + n + 0] = 2; other[n * k + n + 1] = 1; other[n * k + n + 36] = 2
    other[n * k + 7] = 0.25; other[n * k + 2 * n + 7] = 1
    red = bucket + other
    cv = red[n * k + n: n * k + 2 * n].clamp(min=1).sqrt().unsqueeze(1)
    cb = red[n * k + 2 * n:].clamp(min=1).sqrt()
    R = torch.cat([(red[: n * k].view(n, k) / cv).reshape(-1), red[n * k: n * k + n] / cb])
    want_flat, want_base = (base + R) + ((flat - base) - local), base + R  # (an untrained row ends bit-identical to its base)
    flat2, base2 = flat.clone(), base.clone()  # for the fused finish + begin pass below
    torch.cuda.synchronize()
    tr.table_delta_finish(flat.data_ptr(), base.data_ptr(), red.data_ptr(), local.data_ptr(), n, k)
    tr.sync()
    assert torch.allclose(flat, want_flat, atol=1e-6) and torch.allclose(base, want_base, atol=1e-6)
    # finish followed by begin == the fused step kernel
    b_sep, l_sep = torch.empty_like(bucket), torch.empty_like(local)
    tr.table_delta_begin(flat.data_ptr(), base.data_ptr(), n, k, b_sep.data_ptr(), l_sep.data_ptr())
    b_fus, l_fus = torch.empty_like(bucket), torch.empty_like(local)
    tr.table_delta_step(flat2.data_ptr(), base2.data_ptr(), red.data_ptr(), local.data_ptr(), n, k, b_fus.data_ptr(),
                        l_fus.data_ptr())
    tr.sync()
    assert torch.equal(flat2, flat) and torch.equal(base2, base)
    assert torch.equal(b_fus, b_sep) and torch.equal(l_fus, l_sep)
    tr.close()


@pytest.mark.parametrize("k", [5, 8])
def test_fused_table_delta_kernels_align_rule_equal_the_torch_algebra(k):
    """the "align" reconciliation rule (R = S min(1, sum |d_r|^2 / |S|^2), MF's default) as HIP kernels — begin, finish
    and the fused step, through the handle-free entry point on the caller's stream — against ItemTableReplica's torch
    formulation of the same rule, with a hand-made all-reduced bucket of 3 virtual ranks: rows where the ranks agree
    (the mean), where they are orthogonal (the sum), where one rank alone moved (its step) and where nobody did."""
    import torch

    from cornac_amd.dist import RULES, ItemTableReplica

    ds = synth_dataset(50, 40, 300, seed=1)
    tr = _trainer(ds, 4)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    tr.set_stream(st.cuda_stream)
    n = 37
    g = torch.Generator(device="cpu").manual_seed(3)
    base = torch.randn(n * k + n, generator=g).to(dev)
    flat = base.clone()
    rows = torch.tensor([0, 3, 4, 20, 36])
    flat[: n * k].view(n, k)[rows] += torch.randn(len(rows), k, generator=g).to(dev)
    flat[n * k + 7] += 0.5
    flat[n * k + 9] -= 0.25
    host = ItemTableReplica(n, k, torch.device("cpu"), rule="align")  # the torch formulation (the gloo tests' path)
    with torch.cuda.stream(st):
        bucket = torch.empty(n * k + 3 * n, device=dev)
        local = torch.empty(n * k + n, device=dev)
        tr.table_delta_begin(flat.data_ptr(), base.data_ptr(), n, k, bucket.data_ptr(), local.data_ptr(), rule=RULES["align"])
        st.synchronize()
        d = flat - base
        assert torch.equal(bucket[: n * k + n], d) and torch.equal(local, d)
        wV, wB = host._weights(d[: n * k].view(n, k).cpu(), d[n * k:].cpu())
        assert torch.allclose(bucket[n * k + n: n * k + 2 * n].cpu(), wV, rtol=1e-6, atol=0)
        assert torch.allclose(bucket[n * k + 2 * n:].cpu(), wB, rtol=1e-6, atol=0)
        assert float(bucket[n * k + n + 1]) == 0.0 and float(bucket[n * k + n + 3]) > 0.0  # untouched row: weight exactly 0
        # two more ranks: row 0 = two copies of this rank's own delta (aligned: the mean comes out), row 3 = an orthogonal
        # delta (the sum comes out), row 1 = touched by another rank only, bias 7 aligned, bias 9 opposed
        other = torch.zeros_like(bucket)
        oV = other[: n * k].view(n, k)
        dV = d[: n * k].view(n, k)
        oV[0] = 2 * dV[0]
        other[n * k + n + 0] = 2 * (dV[0] * dV[0]).sum()
        orth = torch.zeros(k, device=dev)
        orth[0], orth[1] = -dV[3, 1], dV[3, 0]
        oV[3] = orth
        other[n * k + n + 3] = (orth * orth).sum()
        oV[1] = torch.randn(k, generator=g).to(dev)
        other[n * k + n + 1] = (oV[1] * oV[1]).sum()
        other[n * k + 7] = 0.5; other[n * k + 2 * n + 7] = 0.25
        other[n * k + 9] = 0.5; other[n * k + 2 * n + 9] = 0.25
        red = bucket + other
        fV = host._factors(red[: n * k].view(n, k).cpu(), red[n * k + n: n * k + 2 * n].cpu()).to(dev)
        fB = host._factors(red[n * k: n * k + n].cpu(), red[n * k + 2 * n:].cpu()).to(dev)
        assert abs(float(fV[0]) - 1 / 3) < 1e-5 and abs(float(fV[3]) - 1.0) < 1e-5 and float(fV[1]) == 1.0 and float(fV[2]) == 1.0
        assert abs(float(fB[7]) - 0.5) < 1e-6 and float(fB[9]) == 1.0  # (opposed deltas: the factor is capped at 1)
        R = torch.cat([(red[: n * k].view(n, k) * fV.unsqueeze(1)).reshape(-1), red[n * k: n * k + n] * fB])
        want_flat, want_base = (base + R) + ((flat - base) - local), base + R
        flat2, base2 = flat.clone(), base.clone()
        tr.table_delta_finish(flat.data_ptr(), base.data_ptr(), red.data_ptr(), local.data_ptr(), n, k, rule=RULES["align"])
        st.synchronize()
        assert torch.allclose(flat, want_flat, atol=1e-6) and torch.allclose(base, want_base, atol=1e-6)
        assert torch.equal(flat[2 * k: 3 * k], base[2 * k: 3 * k])  # an untrained row ends bit-identical to its base
        b_sep, l_sep = torch.empty_like(bucket), torch.empty_like(local)
        tr.table_delta_begin(flat.data_ptr(), base.data_ptr(), n, k, b_sep.data_ptr(), l_sep.data_ptr(), rule=RULES["align"])
        b_fus, l_fus = torch.empty_like(bucket), torch.empty_like(local)
        tr.table_delta_step(flat2.data_ptr(), base2.data_ptr(), red.data_ptr(), local.data_ptr(), n, k, b_fus.data_ptr(),
                            l_fus.data_ptr(), rule=RULES["align"])
        st.synchronize()
        assert torch.equal(flat2, flat) and torch.equal(base2, base)
        assert torch.equal(b_fus, b_sep) and torch.equal(l_fus, l_sep)
    tr.close()


def _fresh_port():
    """a free TCP port for this test's rendezvous (several tests of this process create and destroy process groups)"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


@pytest.mark.parametrize("rule", ["sqrt", "align"])
def test_sharded_trainer_keeps_packed_records_between_chunks_through_rccl(rule):
    """The multi-GPU BPR driver over the XCD-strata form (forced here; automatic for item tables of >= 2^20 rows, i.e. the
    configs[4] slices): with the dense exchange the handle keeps its packed item records from chunk to chunk
    (cornac_hip_bpr_chunk_records), the driver's table passes work on the records, and finish() -> sync() writes them
    back into the torch-owned replica.  One rank through RCCL, 4 exchanges per epoch:
    * lr = 0: records -> passes -> records -> replica is the identity, bit for bit (any mis-addressed row or bias would show);
    * lr > 0: the same seed with and without the records trains the same model in aggregate (the sampler and the
      arithmetic are the same; the form itself is racy, so not row by row) and the replica is rebased after finish()."""
    import torch
    import torch.distributed as dist

    from cornac_amd import synth
    from cornac_amd.dist import ShardedBprTrainer

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = _fresh_port()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # 2^20 item rows: the size from which the form is automatic; k = 100: padded rows in the records, scalar table passes
    n_users, n_items, k = 20000, 1 << 20, 100
    users, items = synth.zipf_interactions(n_users, n_items, 700_000, 0.5, 11)   # (>= 524 288: user-row ownership)
    indptr, indices = synth.csr_from_sorted(users, items, n_users)
    nnz = len(indices)
    rs = np.random.RandomState(2)
    U0 = rs.normal(0, 0.1, (n_users, k)).astype(np.float32)
    V0 = rs.normal(0, 0.1, (n_items, k)).astype(np.float32)
    B0 = rs.normal(0, 0.1, n_items).astype(np.float32)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        def drive(lr, records, epochs=1):
            tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
            tr.set_factors(U0, None, None)
            tr.seed_hogwild(21)
            sh = ShardedBprTrainer(tr, n_items, k, dev, sync_every=(nnz + 3) // 4, rule=rule)
            if not records:
                tr.chunk_records(False)
            sh.load_items(V0, B0)
            for _ in range(epochs):
                sh.run(nnz, lr, 0.01, True, _lib.NEG_UNIFORM, _lib.FORM_STRATA)
            c, s_ = sh.finish()
            st = tr.strata_stats()
            V, B = sh.table.V.cpu().numpy(), sh.table.B.cpu().numpy()
            rebased = np.array_equal(sh.table.base.cpu().numpy(), sh.table.flat.cpu().numpy())
            ex = dict(sh.table.exchanges)
            U = tr.get_user_factors()
            tr.close()
            assert st["bucket_builds"] >= 1 and ex["dense"] == 4 * epochs and rebased, (st, ex, rebased)
            return U, V, B, c / float(nnz * epochs - s_)

        U, V, B, _ = drive(0.0, True)
        assert np.array_equal(V, V0) and np.array_equal(B, B0) and np.array_equal(U, U0)
        Ua, Va, Ba, acc_a = drive(0.05, True, epochs=2)
        Ub, Vb, Bb, acc_b = drive(0.05, False, epochs=2)
        assert np.abs(Va - V0).max() > 1e-3 and np.abs(Ba - B0).max() > 1e-3
        # the form is racy (a read-modify-write lost inside an XCD differs from run to run, and its effect spreads through
        # the users that touch the row), so the two tables agree in aggregate, not row by row: the same moves, the same
        # 'correct' fraction
        dva, dvb = (Va - V0).ravel().astype(np.float64), (Vb - V0).ravel().astype(np.float64)
        cos = float(dva @ dvb) / (np.linalg.norm(dva) * np.linalg.norm(dvb))
        dba, dbb = (Ba - B0).astype(np.float64), (Bb - B0).astype(np.float64)
        cos_b = float(dba @ dbb) / (np.linalg.norm(dba) * np.linalg.norm(dbb))
        assert cos > 0.995 and cos_b > 0.995, (cos, cos_b)
        assert abs(np.linalg.norm(dva) / np.linalg.norm(dvb) - 1.0) < 0.01 and abs(acc_a - acc_b) < 0.01, (acc_a, acc_b)
    finally:
        dist.destroy_process_group()


def test_sharded_mf_trainer_single_rank_through_rccl():
    """the multi-GPU MF driver (ShardedMfTrainer: item side [V | Bi] in a torch-owned replica, epochs enqueued in slices on
    the driver's stream, the replica's delta passes through cornac_hip_table_delta, a real RCCL group of size 1) against
    the plain cornac_hip_mf_fit on the same data: with one rank every exchange is a rebase, so both learn the same model
    up to hogwild scheduling — the same loss curve and RMSE within the spread of that; parts_per_epoch = 1 takes the whole-epoch form."""
    import torch
    import torch.distributed as dist

    from cornac_amd import synth
    from cornac_amd.dist import ShardedMfTrainer, global_mean_across_ranks

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = _fresh_port()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_users, n_items, k = 5000, 2000, 64
    users, items = synth.zipf_interactions(n_users, n_items, 600_000, 0.6, 3)
    rs = np.random.RandomState(0)
    P, Q = rs.normal(0, 1, (n_users, 4)), rs.normal(0, 1, (n_items, 4))
    val = np.clip(3.0 + 0.5 * np.einsum("nk,nk->n", P[users], Q[items]) + rs.normal(0, 0.3, len(users)), 1, 5).astype(np.float32)
    U0 = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
    V0 = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
    zu, zi = np.zeros(n_users, np.float32), np.zeros(n_items, np.float32)
    epochs, lr, reg = 6, 0.01, 0.02

    def rmse(U, V, Bu, Bi, mu):
        pred = mu + Bu[users] + Bi[items] + np.einsum("nk,nk->n", U[users], V[items])
        return float(np.sqrt(np.mean((pred - val) ** 2)))

    mu_ref = float(val.astype(np.float64).mean())
    ref = _lib.MfTrainer(users, items, val, n_users, n_items, k)
    ref.set_factors(U0, V0, zu, zi)
    ref.hogwild_form(1)
    loss_ref, _ = ref.fit(epochs, lr, reg, mu_ref, True, False, _lib.MODE_HOGWILD)
    rmse_ref = rmse(*ref.get_factors(), mu_ref)
    ref.close()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        mu = global_mean_across_ranks(val)
        assert abs(mu - mu_ref) < 1e-9
        for parts in (8, 1):
            tr = _lib.MfTrainer(users, items, val, n_users, n_items, k)
            tr.set_factors(U0, None, zu, None)
            sh = ShardedMfTrainer(tr, n_items, k, dev, parts_per_epoch=parts)
            sh.load_items(V0, zi)
            losses = []
            for _ in range(epochs):
                sh.run_epoch(lr, reg, mu)
                losses.append(0.5 * sh.finish())
            U, _, Bu, _ = tr.get_factors()
            V, Bi = sh.table.V.cpu().numpy(), sh.table.B.cpu().numpy()
            assert np.array_equal(sh.table.base.cpu().numpy(), sh.table.flat.cpu().numpy()), "finish() leaves a rebased table"
            assert sh.table.exchanges["dense"] == epochs * parts
            tr.close()
            got = rmse(U, V, Bu, Bi, mu)
            assert np.isfinite(U).all() and np.isfinite(V).all()
            # mid-training comparison (6 epochs): slices of the user-sorted order race a little less than the whole
            # epoch in one launch and sit slightly ahead on the curve (measured 0.454 vs 0.481)
            assert 0.85 * rmse_ref < got < 1.05 * rmse_ref, (parts, got, rmse_ref)
            assert 0.7 * float(loss_ref[-1]) < losses[-1] < 1.1 * float(loss_ref[-1]), (parts, losses, loss_ref)
            assert losses[-1] < losses[0]
    finally:
        dist.destroy_process_group()


def test_model_level_sharded_fits_single_rank_through_rccl():
    """dist.fit_bpr_sharded / fit_mf_sharded (model.fit over a process group) on one rank through a real RCCL group — the
    device path of the drivers they compose: bound item table, driver stream, epochs in exactly N chunks, the size-one
    broadcast / all-gathers — next to the plain model.fit on the same data: both learn the same model up to hogwild
    scheduling (measured: BPR 'correct' 0.7215 vs 0.7206; MF final loss within 7 %)."""
    import torch
    import torch.distributed as dist

    import cornac_amd as ca
    from cornac_amd import synth
    from cornac_amd.dist import fit_bpr_sharded, fit_mf_sharded

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = _fresh_port()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    users, items = synth.zipf_interactions(4000, 1500, 300_000, 0.7, 3)
    rs = np.random.RandomState(0)
    P, Q = rs.normal(0, 1, (4000, 4)), rs.normal(0, 1, (1500, 4))
    val = np.clip(np.rint(3.0 + 0.6 * np.einsum("nk,nk->n", P[users], Q[items]) + rs.normal(0, 0.3, len(users))), 1, 5)
    ds = ca.Dataset.from_uir(list(zip(users.tolist(), items.tolist(), val.tolist())), seed=1)
    nnz = ds.matrix.nnz
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        kw = dict(k=64, max_iter=6, learning_rate=0.05, lambda_reg=0.01, seed=5, mode="hogwild")
        a = fit_bpr_sharded(ca.BPR(**kw), ds, device=dev, sync_per_epoch=8)
        b = ca.BPR(**kw).fit(ds)
        fa = a.fit_stats[0][0] / max(6 * nnz - a.fit_stats[0][1], 1)
        fb = b.fit_stats[0][0] / max(6 * nnz - b.fit_stats[0][1], 1)
        assert np.isfinite(a.u_factors).all() and np.isfinite(a.i_factors).all()
        assert abs(fa - fb) < 0.03 and fa > 0.6, (fa, fb)
        assert np.abs(a.score(3) - (a.i_biases + a.i_factors @ a.u_factors[3])).max() < 1e-5
        with pytest.raises(ValueError):
            fit_bpr_sharded(ca.BPR(k=8, seed=1), ds, device=dev)   # seeded => sequential semantics: refused
        # the same fit as the ring conveyor (regime 2; one rank: two blocks = two halves of the epoch's bins, one handle)
        c = fit_bpr_sharded(ca.BPR(**kw), ds, device=dev, regime="ring")
        fc = c.fit_stats[0][0] / max(6 * nnz - c.fit_stats[0][1], 1)
        assert np.isfinite(c.u_factors).all() and np.isfinite(c.i_factors).all() and abs(fc - fb) < 0.03, (fc, fb)
        assert np.abs(c.score(3) - (c.i_biases + c.i_factors @ c.u_factors[3])).max() < 1e-5
        kw = dict(k=64, max_iter=8, learning_rate=0.01, lambda_reg=0.02, seed=5, mode="hogwild")
        m = fit_mf_sharded(ca.MF(**kw), ds, device=dev, parts_per_epoch=8)
        p = ca.MF(**kw).fit(ds)
        assert m.epochs_run == 8 and m.loss_history[-1] < m.loss_history[0]
        assert abs(m.loss_history[-1] - p.loss_history[-1]) < 0.15 * p.loss_history[-1], (m.loss_history, p.loss_history)
    finally:
        dist.destroy_process_group()


# ---- resident exchange: regime 1 inside ONE launch per epoch (csrc/bpr_ldsbin.inc EXCH kernels, dist.run_epoch_resident) ----
def _ldsbin_problem(k, nnz=300_000, seed=9):
    from cornac_amd import synth

    nu, ni = 4000, 12800
    users, items = synth.zipf_interactions(nu, ni, nnz, 0.6, seed)
    indptr, indices = synth.csr_from_sorted(users, items, nu)
    rs = np.random.RandomState(seed)
    return (nu, ni, indptr, indices, rs.normal(0, 0.1, (nu, k)).astype(np.float32),
            rs.normal(0, 0.1, (ni, k)).astype(np.float32), rs.normal(0, 0.1, ni).astype(np.float32))


@pytest.mark.parametrize("k,n_ex,twin", [(64, 4, True), (64, 16, True), (100, 32, True), (64, 8, False), (128, 5, True)])
def test_resident_exchange_publishes_and_applies_every_row_exactly_once(k, n_ex, twin):
    """The accounting of the resident exchange, made deterministic: lr = 0 (no step changes a row) and a base that differs
    from the table by a known D0, so exchange 0 carries d = D0 for EVERY row — cold rows from their bins' LDS, hot rows
    from the global table — and all later exchanges carry exact zeros.  A twin rank is played on the communication stream
    (bucket *= 2: S = 2 d, two touching ranks), so each row must end at table + (sqrt(2) - 1) D0 with base == table bit
    for bit: a row published twice, never, or corrected twice (in the launch AND by the flush) would show.  Without the
    twin the correction is exactly zero and the table must not change by a bit.  All n_ex arrival counters reach the
    launch's workgroup count, every landed flag is up, the wait never timed out."""
    import torch

    from cornac_amd.dist import ShardedBprTrainer

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    nu, ni, indptr, indices, U0, V0, B0 = _ldsbin_problem(k)
    tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
    try:
        tr.set_factors(U0, None, None)
        tr.seed_hogwild(77)
        sh = ShardedBprTrainer(tr, ni, k, dev, sync_every=len(indices))
        sh.load_items(V0, B0)
        bins = sh.resident_bins()
        assert bins == 256, bins
        rs = np.random.RandomState(k + n_ex)
        D0 = rs.normal(0, 0.05, ni * k + ni).astype(np.float32)
        flat0 = np.concatenate([V0.ravel(), B0])
        base0 = flat0 - D0
        with sh._on_stream():
            sh.table.base.copy_(torch.as_tensor(base0))
        sh.stream.synchronize()
        if twin:
            sh.resident_bucket_hook = lambda e, bucket: bucket.mul_(2.0)
        sh.run_epoch(len(indices), n_ex, 0.0, 0.01, resident=None)
        c, s = sh.finish()
        assert sh.table.exchanges.get("resident") == n_ex, sh.table.exchanges
        sig = sh._resident["signals"].cpu().numpy()
        assert (sig[:n_ex] == bins).all() and (sig[n_ex:2 * n_ex] == 1).all() and sig[2 * n_ex] == 0, sig
        flat, base = sh.table.flat.cpu().numpy(), sh.table.base.cpu().numpy()
        assert np.array_equal(flat, base), "table and base took the same corrections"
        buckets = sh._resident["buckets"].cpu().numpy()
        keeps = sh._resident["keeps"].cpu().numpy()
        d0 = flat0 - base0                                    # the pending delta (fp32, as the kernel computes it)
        # every element's delta is published by exactly ONE exchange — the row's first duty: exchange 0, or a later one
        # where a bin holds so few tiles that two boundaries' duties of one part overtake each other — and every other
        # exchange carries an exact zero for it (row == base bit for bit)
        assert ((keeps != 0).sum(0) <= 1).all() and np.array_equal(keeps.sum(0, dtype=np.float32), d0)
        width = ni * k + ni
        if twin:
            assert np.array_equal(buckets[:, :width], 2.0 * keeps)
            wV, wB = buckets[:, width: width + ni], buckets[:, width + ni:]
            assert np.array_equal(wV, 2.0 * (keeps[:, : ni * k].reshape(n_ex, ni, k) != 0).any(2))
            assert np.array_equal(wB, 2.0 * (keeps[:, ni * k:] != 0))
            f = np.float32(1.0) / np.sqrt(np.float32(2.0))
            want = flat0 + ((2.0 * d0) * f - d0)
            assert np.abs(flat - want).max() < 1e-6, np.abs(flat - want).max()
        else:
            assert np.array_equal(flat, flat0), "one rank alone: the exchange is the identity"
        assert np.array_equal(tr.get_user_factors(), U0) and 0 < s < len(indices) // 4
    finally:
        tr.close()


def test_resident_exchange_trains_like_the_plain_launch_through_rccl():
    """One rank through a real RCCL group: epochs with 16 exchange points inside ONE launch against the plain call on the
    same seed (the LDS-bin sample stream is a pure function of (seed, epoch, bin, draw): both draw the same triplets, the
    skip counters are equal; only the hogwild interleaving differs).  What the protocol promises is checked exactly, per
    epoch: the all-reduced bucket of a single rank is its own delta, and the base moved by exactly the sum of the
    published deltas — i.e. no correction other than zero was ever applied, inside the launch or by the flush.  The
    result is another interleaving of the same updates: it moves like the plain run about as much as the chunk protocol
    does (tools/diag_resident.py: cos of the item-table moves 0.982 / 0.993 / 0.997 for resident / chunks / the plain call
    repeated); the collectives really ran (16 per epoch) and only hot rows hold unpublished steps after an epoch."""
    import torch
    import torch.distributed as dist

    from cornac_amd.dist import ShardedBprTrainer

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = _fresh_port()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    k, epochs = 64, 3
    nu, ni, indptr, indices, U0, V0, B0 = _ldsbin_problem(k, nnz=600_000, seed=4)
    nnz = len(indices)
    plain = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
    plain.set_factors(U0, V0, B0)
    plain.seed_hogwild(5)
    c_p, s_p = plain.fit_epochs(epochs, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    Up, Vp, Bp = plain.get_factors()
    plain.close()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
        tr.set_factors(U0, None, None)
        tr.seed_hogwild(5)
        sh = ShardedBprTrainer(tr, ni, k, dev, sync_every=nnz)
        sh.load_items(V0, B0)
        assert sh.resident_bins() == 256
        width, prev, n_hot = ni * k + ni, np.concatenate([V0.ravel(), B0]), tr.ldsbin_stats()["n_hot"]
        for _ in range(epochs):
            sh.run_epoch(nnz, 16, 0.05, 0.01)
            sh.stream.synchronize()
            sh._resident["comm"].synchronize()
            flat, base = sh.table.flat.cpu().numpy(), sh.table.base.cpu().numpy()
            buckets, keeps = sh._resident["buckets"].cpu().numpy(), sh._resident["keeps"].cpu().numpy()
            assert np.array_equal(buckets[:, :width], keeps), "one rank: the all-reduced delta is its own"
            assert np.abs((base - prev) - keeps.astype(np.float64).sum(0)).max() < 1e-6, "a non-zero correction was applied"
            rows_left = np.unique(np.nonzero((flat != base)[: ni * k])[0] // k)
            assert len(rows_left) <= n_hot, (len(rows_left), n_hot)   # only hot rows move after their last publication
            prev = base
        c, s = sh.finish()
        V, B, U = sh.table.V.cpu().numpy(), sh.table.B.cpu().numpy(), tr.get_user_factors()
        tr.close()
        assert sh.table.exchanges["resident"] == 16 * epochs and s == s_p, (sh.table.exchanges, s, s_p)
        assert abs(c - c_p) < 0.005 * nnz * epochs, (c, c_p)
        for got, ref, init in ((V, Vp, V0), (B, Bp, B0), (U, Up, U0)):
            dg, dr = (got - init).ravel().astype(np.float64), (ref - init).ravel().astype(np.float64)
            cos = float(dg @ dr) / (np.linalg.norm(dg) * np.linalg.norm(dr))
            assert cos > 0.97 and abs(np.linalg.norm(dg) / np.linalg.norm(dr) - 1.0) < 0.02, cos
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("emulate,rings,virtual", [(False, 1, None), (True, 1, None), (True, 2, None), (False, 4, 8)])
def test_ring_conveyor_single_rank_on_the_device(emulate, rings, virtual):
    """BinConveyorBprTrainer (multi-GPU regime 2) on ONE rank with the real handle in conveyor layout: the blocks are bin ranges
    of the epoch's deal, a step is ONE launch over `rings` bin ranges reading / writing its rows in the block buffers
    (cornac_hip_bpr_conveyor_enqueue), the rows are re-dealt to new slots between the epochs; with emulate_traffic the trained
    block is copied to the free buffer on the communication stream, as a neighbour's receive would.
    (1) lr = 0 returns every table bit for bit — through the epochs' re-deals — and draws nnz samples per epoch; the skip
    counter equals the oracle's restatement of the sampler under the conveyor's deal key, draw for draw;
    (2) reg = 0 conserves the column sums of the item table (every step adds +d to the positive's row and -d to the
    negative's: exact in LDS, so only fp32 rounding moves the sum);
    (3) training moves every block, stays finite, and learns like the plain single-handle fit of the same data.
    virtual = 8 with four rings: the layout of an 8-rank node (64 blocks, 16 steps per epoch, four ranges per launch)."""
    import torch

    from cornac_amd import synth
    from cornac_amd.dist import BinConveyorBprTrainer
    from oracle import oracle as orc

    n_users, n_items, k = 30000, 4001, 64
    users, items = synth.zipf_interactions(n_users, n_items, 1_200_000, 0.7, 4)
    indptr, indices = synth.csr_from_sorted(users, items, n_users)
    nnz = len(indices)
    rs = np.random.RandomState(0)
    U = ((rs.uniform(0, 1, (n_users, k)) - 0.5) / k).astype(np.float32)
    V = ((rs.uniform(0, 1, (n_items + 7, k)) - 0.5) / k).astype(np.float32)   # 7 rows beyond the train items: never touched
    B = rs.normal(0, 0.01, n_items + 7).astype(np.float32)
    dev = torch.device("cuda", 0)
    ring = BinConveyorBprTrainer(indptr, indices, n_users, n_items + 7, k, dev, seed=3, emulate_traffic=emulate, rings=rings,
                                 n_train_items=n_items, virtual_world=virtual)
    nb = 2 * (virtual or 1)
    assert ring.nb == nb and ring.K == rings and ring.nnz == nnz and ring.n_bins == ring.bpb * nb * rings
    assert ring.cap * ring.n_bins >= n_items and ring.cap >= 16
    ring.set_user_factors(U)
    ring.load_items(V, B)
    for _ in range(3):
        ring.run_epoch(0.0, 0.0)
    c, s = ring.finish()
    V1, B1 = ring.gather()
    assert ring.redeals == 2 and 0 < s < 0.2 * 3 * nnz and c + s <= 3 * nnz
    assert np.array_equal(V1, V) and np.array_equal(B1, B) and np.array_equal(ring.get_user_factors(), U)
    tables = orc.ldsbin_tables(indptr, indices, n_items, ring.n_bins, 10 ** 9)      # (no hot items in the conveyor layout)
    hog_seed = (3 * 0x9E3779B97F4A7C15 + 1) & 0xFFFFFFFFFFFFFFFF
    want = sum(orc.ldsbin_epoch(hog_seed, e, ring.n_bins, 10 ** 9, indptr, indices, n_items, tables=tables,
                                deal=(ring.deal_seed, e))[0] for e in range(3))
    assert s == want, "the conveyor's sampler deviates from its oracle restatement: %d != %d" % (s, want)
    assert ring.trainer.tr.ldsbin_stats()["lock_timeouts"] == 0
    # (2) reg = 0: column sums
    ring.run_epoch(0.05, 0.0)
    ring.finish()
    Vc, Bc = ring.gather()
    assert np.abs(Vc - V).max() > 1e-4
    drift = np.abs(Vc[:n_items].astype(np.float64).sum(0) - V[:n_items].astype(np.float64).sum(0)).max()
    assert drift < 2e-3 and abs(float(Bc[:n_items].astype(np.float64).sum() - B[:n_items].astype(np.float64).sum())) < 2e-3, drift
    # (3) training
    ring.set_user_factors(U)
    ring.load_items(V, B)
    for _ in range(5):
        ring.run_epoch(0.05, 0.01)
    ring.finish()
    ring.run_epoch(0.05, 0.01)
    c, s = ring.finish()
    acc_ring = c / (nnz - s)
    V2, B2 = ring.gather()
    U2 = ring.get_user_factors()
    assert ring.steps_trained[: nb * rings] == [(t, t * rings + g) for t in range(nb) for g in range(rings)]
    ring.close()
    assert np.isfinite(V2).all() and np.isfinite(U2).all() and np.array_equal(V2[n_items:], V[n_items:])
    assert (np.abs(V2[:n_items] - V[:n_items]).max(1) > 1e-4).mean() > 0.99 and np.abs(U2 - U).max() > 1e-3
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    tr.set_factors(U, V[:n_items], B[:n_items])
    tr.seed_hogwild(3)
    tr.fit_epochs(5, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    c, s = tr.fit_epochs(1, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    tr.close()
    assert abs(acc_ring - c / (nnz - s)) < 0.02, (acc_ring, c / (nnz - s))


def _planted_slice(rank, n_users, n_items, degree, n_clusters):
    """a rank's user slice at the configs[4] density pattern (a handful of items per user, Zipf-free) with structure to
    learn: user u prefers the items of its taste cluster (item i in cluster i % n_clusters) for 4 of its 5 items"""
    rs = np.random.RandomState(500 + rank)
    c = rs.randint(0, n_clusters, size=n_users)
    own = rs.randint(0, n_items // n_clusters, size=(n_users, degree)) * n_clusters + c[:, None]
    anyi = rs.randint(0, n_items, size=(n_users, degree))
    items = np.where(np.arange(degree)[None, :] < degree - 1, own, anyi)
    items.sort(axis=1)
    keep = np.ones_like(items, bool)
    keep[:, 1:] = items[:, 1:] != items[:, :-1]
    indptr = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int32)
    return indptr, items[keep].astype(np.int32)


def test_eight_virtual_ranks_at_the_slice_density_ring_conveyor_against_replicas(capsys):
    """The two multi-GPU regimes at the configs[4] density (6.25 draws per item row, rank and epoch), R = 8 virtual ranks ON
    THE DEVICE, each its own user slice, next to ONE process that trains all eight slices' users together; the measure is
    the pairwise accuracy on rank 0's probe triplets in the MIDDLE of training, where a stale item side costs most.

      * regime 2, the ring conveyor (BinConveyorBprTrainer's schedule: 16 blocks = bin ranges of the epoch's deal, rank r
        trains block (2 r + t) % 16 in step t, re-dealt between the epochs; here executed step by step on one device — the
        serial execution the gloo test shows the ranks' parallel run to equal): every item row is in one place, nothing is
        reconciled.  Gate: within 1.5 points of the single process.
      * regime 1, replicas reconciled with ItemTableReplica's "align" algebra once every 1 / 2 / 4 epochs (round 4's
        exchange_schedule returned 4 here on the strength of a CPU toy; since round 6 it returns 1): every rank moves every item row the same way, the
        rule averages the R aligned deltas, and the shared item side learns at a fraction of the single process's pace —
        measured here, printed, and the reason the conveyor is the regime for this shape.  Only at convergence do the
        replicas catch up (gate: within 1.5 points after twice the epochs)."""
    import torch

    from cornac_amd.dist import exchange_schedule

    R, n_items, k, epochs, lr, reg, C = 8, 48_000, 32, 16, 0.1, 0.01, 60
    n_users, degree = 60_000, 5
    slices = [_planted_slice(r, n_users, n_items, degree, C) for r in range(R)]
    nnz_r = max(len(ix) for _, ix in slices)
    rs = np.random.RandomState(0)
    V0 = ((rs.uniform(0, 1, (n_items, k)) - .5) / k).astype(np.float32)
    U0 = [((np.random.RandomState(10 + r).uniform(0, 1, (n_users, k)) - .5) / k).astype(np.float32) for r in range(R)]
    prs = np.random.RandomState(99)
    ip0, ix0 = slices[0]
    pp = prs.randint(0, len(ix0), 100_000)
    probe_u = np.repeat(np.arange(n_users), np.diff(ip0))[pp]
    probe_i, probe_j = ix0[pp], prs.randint(0, n_items, len(pp))

    def accuracy(U, V, B):
        sc = np.einsum("nk,nk->n", U[probe_u], V[probe_i] - V[probe_j]) + B[probe_i] - B[probe_j]
        return float((sc > 0).mean())

    def one_process(n_epochs):
        ip_all = np.concatenate([[0]] + [ip[1:].astype(np.int64) + sum(len(s[1]) for s in slices[:r]) for r, (ip, _) in enumerate(slices)]).astype(np.int32)
        tr = _lib.BprTrainer(ip_all, np.concatenate([ix for _, ix in slices]), R * n_users, n_items, R * n_users, n_items, k)
        tr.set_factors(np.concatenate(U0), V0, np.zeros(n_items, np.float32))
        tr.seed_hogwild(5)
        tr.fit_epochs(n_epochs, lr, reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
        U1, V1, B1 = tr.get_factors()
        tr.close()
        return accuracy(U1[:n_users], V1, B1)

    def replicas(every, n_epochs):
        trainers = []
        for r, (ip, ix) in enumerate(slices):
            t = _lib.BprTrainer(ip, ix, n_users, n_items, n_users, n_items, k)
            t.set_factors(U0[r], V0, np.zeros(n_items, np.float32))
            t.seed_hogwild(1000 + r)
            trainers.append(t)
        V, B = V0.copy(), np.zeros(n_items, np.float32)
        for e0 in range(0, n_epochs, every):
            dV, dB = np.zeros_like(V), np.zeros_like(B)
            qV, qB = np.zeros(n_items, np.float64), np.zeros(n_items, np.float64)
            for t in trainers:
                t.set_factors(None, V, B)
                t.fit_epochs(min(every, n_epochs - e0), lr, reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
                Vr, Br = t.get_item_factors()
                dV += Vr - V; dB += Br - B
                qV += ((Vr - V).astype(np.float64) ** 2).sum(1); qB += (Br - B).astype(np.float64) ** 2
            nV, nB = (dV.astype(np.float64) ** 2).sum(1), dB.astype(np.float64) ** 2   # rule "align" (ItemTableReplica._factors)
            dV *= np.where(nV > 0, np.minimum(1.0, qV / np.maximum(nV, 1e-300)), 1.0).astype(np.float32)[:, None]
            dB *= np.where(nB > 0, np.minimum(1.0, qB / np.maximum(nB, 1e-300)), 1.0).astype(np.float32)
            V, B = V + dV, B + dB
        Ur = trainers[0].get_user_factors()
        for t in trainers:
            t.close()
        return accuracy(Ur, V, B)

    def conveyor(n_epochs):
        """the conveyor's schedule executed step by step on one device (the serial execution the gloo test shows the ranks'
        parallel run to equal): one handle per virtual rank in conveyor layout over 16 blocks, ONE set of block buffers, the rows
        re-dealt to the next epoch's slots between the epochs"""
        dev, nb = torch.device("cuda", 0), 2 * R
        deg = sum(np.bincount(ix, minlength=n_items) for _, ix in slices)
        order = np.argsort(-deg, kind="stable").astype(np.int32)
        Us = [torch.as_tensor(U0[r]).to(dev) for r in range(R)]
        torch.cuda.synchronize()
        handles = []
        for r, (ip, ix) in enumerate(slices):
            t = _lib.BprTrainer(ip, ix, n_users, n_items, n_users, n_items, k)
            t.bind_device(Us[r].data_ptr(), None, None)
            t.seed_hogwild(7000 + 100 * r)
            dims = t.conveyor_setup(nb, order, 4242)
            handles.append(t)
        n_bins, bpb, cap = dims
        W = bpb * cap
        table = torch.zeros((nb, W * (k + 1)), dtype=torch.float32, device=dev)     # the 16 block buffers

        def layout(e):
            si = torch.empty(n_bins * cap, dtype=torch.int32, device=dev)
            handles[0].conveyor_layout(e, si.data_ptr(), None)
            handles[0].sync()
            return si.long()

        def scatter(V, B, si):
            ok = si >= 0
            rows = torch.zeros((nb * W, k), device=dev)
            bias = torch.zeros(nb * W, device=dev)
            rows[ok], bias[ok] = V[si[ok]], B[si[ok]]
            table[:, : W * k] = rows.view(nb, W * k)
            table[:, W * k:] = bias.view(nb, W)

        def collect(si):
            ok = si >= 0
            V, B = torch.zeros((n_items, k), device=dev), torch.zeros(n_items, device=dev)
            V[si[ok]] = table[:, : W * k].reshape(nb * W, k)[ok]
            B[si[ok]] = table[:, W * k:].reshape(nb * W)[ok]
            return V, B

        si = layout(0)
        scatter(torch.as_tensor(V0).to(dev), torch.zeros(n_items, device=dev), si)
        torch.cuda.synchronize()
        for e in range(n_epochs):
            if e:
                V, B = collect(si)
                si = layout(e)
                scatter(V, B, si)
                torch.cuda.synchronize()
            for step in range(nb):
                for r in range(R):
                    blk = (2 * r + step) % nb
                    handles[r].conveyor_enqueue(e, e, [blk], [table[blk].data_ptr()], lr, reg, True, _lib.NEG_UNIFORM, 0)
                    handles[r].sync()     # (another handle trains this block in the next step)
        V, B = collect(si)
        assert sum(t.ldsbin_stats()["lock_timeouts"] for t in handles) == 0
        for t in handles:
            t.close()
        return accuracy(Us[0].cpu().numpy(), V.cpu().numpy(), B.cpu().numpy())

    acc_one = one_process(epochs)
    acc_ring = conveyor(epochs)
    acc_rep = {every: replicas(every, epochs) for every in (1, 2, 4)}
    late = (one_process(2 * epochs), replicas(exchange_schedule(nnz_r, n_items)[1], 2 * epochs))
    with capsys.disabled():
        print("\n8 virtual ranks at the configs[4] density (%d items, %d draws per rank and epoch), pairwise accuracy on rank 0's probe "
              "after %d epochs: one process on all data %.4f | ring conveyor of 16 item blocks %.4f | replicas, 'align', one exchange "
              "every 1 / 2 / 4 epochs: %.4f / %.4f / %.4f | after %d epochs: one process %.4f, replicas at exchange_schedule's interval %.4f"
              % (n_items, nnz_r, epochs, acc_one, acc_ring, acc_rep[1], acc_rep[2], acc_rep[4], 2 * epochs, late[0], late[1]))
    assert acc_one > 0.9, acc_one
    assert acc_ring >= acc_one - 0.015, (acc_ring, acc_one)
    assert acc_rep[1] >= acc_rep[2] - 0.01 >= acc_rep[4] - 0.02, acc_rep       # staler replicas never help
    assert late[1] >= late[0] - 0.015, late


def test_negative_population_of_the_whole_matrix_on_a_user_slice():
    """WBPR over several ranks (recom_wbpr.pyx:135: the negative is the item of a uniformly drawn interaction of the WHOLE
    matrix): a handle that holds a user slice is given the global population (cornac_hip_bpr_set_negative_population);
    the sampled negatives then follow the GLOBAL item degrees, not the slice's — and a WBPR epoch still trains (fused
    kernel: the LDS-bin form's binned popularity draw weights by the handle's own interactions)."""
    import torch

    from cornac_amd import synth
    from cornac_amd.dist import global_negative_population

    n_users, n_items, k = 6000, 3003, 32
    users, items = synth.zipf_interactions(n_users, n_items, 600_000, 0.9, 5)
    indptr, indices = synth.csr_from_sorted(users, items, n_users)
    # the slice: the first 1500 users, whose own interactions are made UNLIKE the global ones: only even items
    m = (users < 1500) & (items % 2 == 0)
    ip, ix = synth.csr_from_sorted(users[m], items[m], 1500)
    pop = global_negative_population(indices, n_items)             # (one process: the whole matrix's degrees)
    assert len(pop) == len(indices) and np.array_equal(np.bincount(pop, minlength=n_items), np.bincount(indices, minlength=n_items))
    tr = _lib.BprTrainer(ip, ix, 1500, n_items, 1500, n_items, k)
    tr.seed_hogwild(21)
    dev = torch.device("cuda", 0)
    n = 400_000
    u, i, j = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))

    def neg_hist():
        torch.cuda.synchronize()
        tr.sample_triplets(n, u.data_ptr(), i.data_ptr(), j.data_ptr(), _lib.NEG_POPULARITY)
        tr.sync()
        jj = j.cpu().numpy()
        return np.bincount(jj[jj >= 0], minlength=n_items).astype(np.float64)

    own = neg_hist()
    assert own[1::2].sum() == 0, "by default the population is the handle's own interactions (even items only here)"
    tr.set_negative_population(pop)
    glob = neg_hist()
    assert glob[1::2].sum() > 0.3 * glob.sum()
    want = np.bincount(indices, minlength=n_items).astype(np.float64)
    top = np.argsort(-want)[:200]
    # accepted negatives ~ global degree x P(not a positive of the drawn user): compare the popular head up to that factor
    ratio = (glob[top] / glob.sum()) / (want[top] / want.sum())
    assert 0.5 < np.median(ratio) < 1.2 and np.corrcoef(glob[top], want[top])[0, 1] > 0.5, (np.median(ratio),)
    rs = np.random.RandomState(0)
    U = ((rs.uniform(0, 1, (1500, k)) - .5) / k).astype(np.float32)
    V = ((rs.uniform(0, 1, (n_items, k)) - .5) / k).astype(np.float32)
    tr.set_factors(U, V, np.zeros(n_items, np.float32))
    c1, s1 = tr.fit_epochs(1, 0.05, 0.01, True, _lib.NEG_POPULARITY, _lib.MODE_HOGWILD)
    c2, s2 = tr.fit_epochs(3, 0.05, 0.01, True, _lib.NEG_POPULARITY, _lib.MODE_HOGWILD)
    U2, V2, _ = tr.get_factors()
    assert np.isfinite(V2).all() and np.abs(V2[1::2] - V[1::2]).max() > 1e-4, "odd items are negatives only through the global population"
    assert c2 / (3 * len(ix) - s2) > c1 / (len(ix) - s1) - 0.01
    tr.set_negative_population(None)
    assert neg_hist()[1::2].sum() == 0
    tr.close()


def test_eight_virtual_ranks_mf_block_rotation_and_replicas_against_one_process(capsys):
    """MF over R = 8 virtual ranks ON THE DEVICE at the Netflix density (~200 ratings per user, thousands per item), every
    rank its own users, next to ONE process that trains all ratings; the measure is the RMSE on HELD-OUT ratings in the
    MIDDLE of training, where a stale item side costs most.

      * regime 2, block rotation (MfBlockRotationTrainer's schedule: 16 item blocks, rank r trains block (2 r + t) % 16 in
        step t; here executed step by step on one device with one cornac_hip_mf handle per (rank, block), all bound to ONE set
        of block buffers — the serial execution the gloo test shows the ranks' parallel run to equal bit for bit): every rating
        is applied exactly once per epoch to the one copy of its item row.  Gate: held-out RMSE not more than 0.5 % above one process's.
      * regime 1, replicas of the item side reconciled with ItemTableReplica's "align" algebra 16 times per epoch
        (ShardedMfTrainer's count at R = 8): measured and printed — 0.857 where one process has 0.531 after 4 epochs: the shared
        rows move at a fraction of the pace (the same finding as for BPR's replicas), which is why fit_mf_sharded(regime="auto")
        takes the rotation."""
    import torch

    from cornac_amd.dist import _DeviceMfBlockTrainer

    R, nu_r, n_items, k, per_user, mid, lr, reg = 8, 6000, 4800, 64, 200, 4, 0.01, 0.02
    rs = np.random.RandomState(0)
    q = rs.normal(0, 1, (n_items, 6)).astype(np.float32)
    ib = rs.normal(0, 0.4, n_items).astype(np.float32)
    pop = 1.0 / np.arange(1, n_items + 1) ** 0.45
    pop /= pop.sum()
    data = []
    for r in range(R):
        rr = np.random.RandomState(100 + r)
        cid = rr.choice(n_items, (nu_r, per_user), p=pop).astype(np.int64)      # (duplicates within a user are fine for SGD)
        rid = np.repeat(np.arange(nu_r, dtype=np.int64), per_user)
        taste = rr.normal(0, 1, (nu_r, 6)).astype(np.float32)
        val = 3.2 + ib[cid] + 0.45 * np.einsum("uf,unf->un", taste, q[cid]) + rr.normal(0, 0.5, cid.shape)
        val = np.clip(val, 1, 5).astype(np.float32).reshape(-1)
        cid = cid.reshape(-1)
        hold = rr.uniform(size=len(val)) < 0.1
        data.append((rid[~hold], cid[~hold], val[~hold], rid[hold], cid[hold], val[hold]))
    mu = float(np.mean(np.concatenate([d[2] for d in data])))
    init = np.random.RandomState(7)
    V0 = init.normal(0, 0.01, (n_items, k)).astype(np.float32)
    U0 = [np.random.RandomState(50 + r).normal(0, 0.01, (nu_r, k)).astype(np.float32) for r in range(R)]

    def rmse(Us, Bus, V, Bi):
        se, n = 0.0, 0
        for r in range(R):
            _, _, _, hr, hc, hv = data[r]
            pred = mu + Bus[r][hr] + Bi[hc] + np.einsum("nk,nk->n", Us[r][hr], V[hc])
            se += float(((pred - hv) ** 2).sum())
            n += len(hv)
        return float(np.sqrt(se / n))

    def one_process(epochs):
        rid = np.concatenate([d[0] + r * nu_r for r, d in enumerate(data)])
        cid, val = np.concatenate([d[1] for d in data]), np.concatenate([d[2] for d in data])
        tr = _lib.MfTrainer(rid, cid, val, R * nu_r, n_items, k)
        tr.set_factors(np.concatenate(U0), V0, np.zeros(R * nu_r, np.float32), np.zeros(n_items, np.float32))
        tr.fit(epochs, lr, reg, mu, True, False, _lib.MODE_HOGWILD)
        U, V, Bu, Bi = tr.get_factors()
        tr.close()
        return rmse([U[r * nu_r:(r + 1) * nu_r] for r in range(R)], [Bu[r * nu_r:(r + 1) * nu_r] for r in range(R)], V, Bi)

    def rotation(epochs):
        dev, nb = torch.device("cuda", 0), 2 * R
        pos = np.empty(n_items, np.int64)
        pos[np.argsort(-np.bincount(np.concatenate([d[1] for d in data]), minlength=n_items), kind="stable")] = np.arange(n_items)
        blk, row, W = pos % nb, pos // nb, (n_items + nb - 1) // nb
        stream = torch.cuda.Stream(dev)
        bufs = [torch.zeros(W * (k + 1), device=dev) for _ in range(nb)]
        for b in range(nb):
            items = np.flatnonzero(blk == b)
            bufs[b][: W * k].view(W, k).index_copy_(0, torch.as_tensor(row[items], device=dev), torch.as_tensor(V0[items]).to(dev))
        Us = [torch.as_tensor(U0[r]).to(dev) for r in range(R)]
        Bus = [torch.zeros(nu_r, device=dev) for _ in range(R)]
        torch.cuda.synchronize()
        handles = []
        for r in range(R):
            rid, cid, val = data[r][:3]
            hs = []
            for b in range(nb):
                sel = np.flatnonzero(blk[cid] == b)
                hs.append(_DeviceMfBlockTrainer(rid[sel], row[cid[sel]], val[sel], nu_r, W, k, Us[r], Bus[r], stream, 0))
            handles.append(hs)
        with torch.cuda.stream(stream):
            for _ in range(epochs):
                for t in range(nb):
                    for r in range(R):
                        b = (2 * r + t) % nb
                        handles[r][b].enqueue(bufs[b][: W * k].view(W, k), bufs[b][W * k:], lr, reg, mu, True)
        for hs in handles:
            for h in hs:
                h.sync()
                h.close()
        V, Bi = np.zeros((n_items, k), np.float32), np.zeros(n_items, np.float32)
        for b in range(nb):
            items = np.flatnonzero(blk == b)
            flat = bufs[b].cpu().numpy()
            V[items], Bi[items] = flat[: W * k].reshape(W, k)[row[items]], flat[W * k:][row[items]]
        return rmse([u.cpu().numpy() for u in Us], [b_.cpu().numpy() for b_ in Bus], V, Bi)

    def replicas(epochs, parts=16):
        trainers = []
        for r in range(R):
            t = _lib.MfTrainer(*data[r][:3], nu_r, n_items, k)
            t.set_factors(U0[r], V0, np.zeros(nu_r, np.float32), np.zeros(n_items, np.float32))
            trainers.append(t)
        V, Bi = V0.copy(), np.zeros(n_items, np.float32)
        for _ in range(epochs):
            for p in range(parts):
                dV, dB = np.zeros_like(V), np.zeros_like(Bi)
                qV, qB = np.zeros(n_items, np.float64), np.zeros(n_items, np.float64)
                for t in trainers:
                    t.set_factors(None, V, None, Bi)
                    t.epoch_enqueue(p, parts, lr, reg, mu, True)
                    t.sync()
                    _, Vr, _, Br = t.get_factors()
                    dV += Vr - V; dB += Br - Bi
                    qV += ((Vr - V).astype(np.float64) ** 2).sum(1); qB += (Br - Bi).astype(np.float64) ** 2
                nV, nB = (dV.astype(np.float64) ** 2).sum(1), dB.astype(np.float64) ** 2   # rule "align" (ItemTableReplica._factors)
                dV *= np.where(nV > 0, np.minimum(1.0, qV / np.maximum(nV, 1e-300)), 1.0).astype(np.float32)[:, None]
                dB *= np.where(nB > 0, np.minimum(1.0, qB / np.maximum(nB, 1e-300)), 1.0).astype(np.float32)
                V, Bi = V + dV, Bi + dB
        Us, Bus = [], []
        for t in trainers:
            U, _, Bu, _ = t.get_factors()
            Us.append(U); Bus.append(Bu)
            t.close()
        return rmse(Us, Bus, V, Bi)

    start = rmse(U0, [np.zeros(nu_r, np.float32)] * R, V0, np.zeros(n_items, np.float32))
    one, rot, rep = one_process(mid), rotation(mid), replicas(mid)
    one_late, rot_late = one_process(3 * mid), rotation(3 * mid)
    with capsys.disabled():
        print("\n8 virtual MF ranks at the Netflix density (%d ratings, %d per item), held-out RMSE after %d epochs (start %.4f): one "
              "process %.4f | block rotation over 16 item blocks %.4f | replicas, 'align', 16 exchanges per epoch %.4f | after %d "
              "epochs: one process %.4f, block rotation %.4f"
              % (sum(len(d[2]) for d in data), sum(len(d[2]) for d in data) // n_items, mid, start, one, rot, rep, 3 * mid, one_late, rot_late))
    assert one < 0.9 * start                              # mid-training: the model has learnt, and is still learning
    assert one_late < one
    # not worse than one process by more than 0.5 % (measured: ~0.9 % BETTER mid-training — every rating lands on the current
    # copy of its rows, where one process's atomic kernel works with a few stale ones), nor different in kind
    assert 0.97 * one <= rot <= 1.005 * one, (rot, one)
    assert 0.97 * one_late <= rot_late <= 1.005 * one_late, (rot_late, one_late)
    assert rep < start, (rep, start)     # the replicas learn — at a fraction of the pace (printed above; why fit_mf_sharded rotates)
