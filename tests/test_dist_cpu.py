"""world_size-2 gloo tests (CPU) of the multi-GPU path's host logic (cornac_amd/dist.py): user
partitioning and the item-table delta exchange (sum of the deltas / sqrt of the number of touching ranks, per row).  The HIP trainer is replaced by a host stand-in
that applies known per-rank updates, so the reduction algebra is checked exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cornac_amd.dist import ItemTableReplica, ShardedBprTrainer, partition_users_by_nnz, slice_csr


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeTrainer:
    """stands in for _lib.BprTrainer on a CPU host: every enqueue adds rank-dependent deltas to V/B"""

    def __init__(self, table, rank):
        self.table, self.rank, self.calls = table, rank, []

    def hogwild_enqueue(self, n, lr, reg, use_bias, neg_population, flags):
        self.calls.append(n)
        self.table.V[self.rank::2] += lr * n  # each rank touches its own stripe of rows ...
        self.table.V[0] += 1.0                # ... and both touch row 0
        self.table.B.add_((self.rank + 1) * 0.5)

    def sync(self):
        return (sum(self.calls), 0)


def _worker(rank, world, port, out, sparse_threshold=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        sh = ShardedBprTrainer(None, total_items=6, k=4, device=dev, sync_every=100, sparse_threshold=sparse_threshold)
        sh.trainer = _FakeTrainer(sh.table, rank)
        V0 = np.arange(24, dtype=np.float32).reshape(6, 4)
        sh.load_items(V0, np.zeros(6, np.float32))
        sh.run(250, lr=0.01, reg=0.0)  # chunks of 100, 100, 50 -> 3 syncs
        correct, _ = sh.finish()
        out[rank] = (sh.table.V.numpy().copy(), sh.table.B.numpy().copy(), sh.trainer.calls, correct)
        out["exchanges%d" % rank] = dict(sh.table.exchanges)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sparse_threshold", [None, 1.0])
def test_item_table_allreduce_of_deltas_world2(sparse_threshold):
    """dense form (one all-reduced bucket) and sparse form (all_gather of the touched rows' records): the same
    reconciliation rule, the same table"""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out, sparse_threshold), nprocs=2, join=True)
    (V_a, B_a, calls_a, c_a), (V_b, B_b, calls_b, c_b) = out[0], out[1]
    ex = out["exchanges0"]
    assert (ex["sparse"], ex["dense"]) == ((3, 0) if sparse_threshold else (0, 3)), ex
    assert calls_a == calls_b == [100, 100, 50] and c_a == 250
    assert np.array_equal(V_a, V_b) and np.array_equal(B_a, B_b), "replicas must agree after every sync"
    V0 = np.arange(24, dtype=np.float32).reshape(6, 4)
    want = V0.copy()
    want[0::2] += 0.01 * 250  # rank 0's stripe: rows only rank 0 touched keep its steps unchanged
    want[1::2] += 0.01 * 250  # rank 1's stripe
    # row 0 is touched by BOTH ranks in every chunk (rank 0: lr*n + 1, rank 1: + 1): summed delta / sqrt(2)
    want[0] = V0[0] + sum((0.01 * n + 2.0) / np.sqrt(2.0) for n in (100, 100, 50))
    assert np.allclose(V_a, want, atol=1e-5)
    assert np.allclose(B_a, 3 * (0.5 + 1.0) / np.sqrt(2.0))  # every bias touched by both ranks


def _align_worker(rank, world, port, out, sparse_threshold):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = ItemTableReplica(5, 2, torch.device("cpu"), sparse_threshold=sparse_threshold, rule="align")
        t.load(np.zeros((5, 2), np.float32), np.zeros(5, np.float32))
        t.V[0] += torch.tensor([1.0, 0.0])                       # both ranks make the SAME step on row 0 ...
        t.V[1] += torch.tensor([1.0, 0.0]) if rank == 0 else torch.tensor([0.0, 2.0])  # ... orthogonal steps on row 1
        if rank == 1:
            t.V[2] += torch.tensor([0.5, 0.5])                   # row 2: rank 1 alone; rows 3, 4: nobody
        t.B[0] += 0.5 if rank == 0 else 1.5                      # biases: same direction, different sizes
        t.B[1] += 1.0 if rank == 0 else -1.0                     # opposed and equal: they cancel
        t.sync()
        out[rank] = (t.V.numpy().copy(), t.B.numpy().copy(), t.base.numpy().copy(), dict(t.exchanges))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sparse_threshold", [None, 1.0])
def test_align_rule_world2(sparse_threshold):
    """rule "align": R = S min(1, sum |d_r|^2 / |S|^2) — identical steps are averaged, orthogonal steps summed, a row one
    rank touched keeps that rank's step; the dense bucket and the sparse records give the same table"""
    out = mp.Manager().dict()
    mp.spawn(_align_worker, args=(2, _free_port(), out, sparse_threshold), nprocs=2, join=True)
    (V0, B0, base0, ex0), (V1, B1, base1, ex1) = out[0], out[1]
    assert (ex0["sparse"], ex0["dense"]) == ((1, 0) if sparse_threshold else (0, 1))
    assert np.array_equal(V0, V1) and np.array_equal(B0, B1) and np.array_equal(base0, base1)
    assert np.allclose(V0[0], [1.0, 0.0])        # S = (2, 0), sum |d|^2 = 2, |S|^2 = 4: factor 1/2 = the mean
    assert np.allclose(V0[1], [1.0, 2.0])        # orthogonal: |S|^2 = 5 = sum |d|^2: the plain sum
    assert np.allclose(V0[2], [0.5, 0.5])        # one rank: its step
    assert np.array_equal(V0[3:], np.zeros((2, 2), np.float32))
    assert np.allclose(B0[0], 2.0 * (0.25 + 2.25) / 4.0)   # S = 2, sum d^2 = 2.5: factor 0.625
    assert B0[1] == 0.0 and np.array_equal(base0, np.concatenate([V0.ravel(), B0]))


def test_align_rule_is_stable_where_the_sqrt_rule_diverges():
    """the advisor's round-3 case, emulated on the host: R = 8 ranks each hold their own ratings of ONE hot item (and
    their own user rows), train a slice with MF's squared-error SGD steps and exchange the item row with the replica's
    algebra, remote deltas one slice late.  Every rank nearly solves the row locally, so 8 deltas summed / sqrt(8)
    overshoot by ~2.8x per exchange and the row oscillates out of range; the align rule averages the (aligned) deltas
    and converges to the value one process reaches."""
    def run(rule, R=8, n=400, exchanges=12, lr=0.05):
        rs = np.random.RandomState(0)
        target = 2.0                                   # every rank's ratings say: item bias = 2
        b = np.zeros(R)                                # the replicas of the hot item's bias
        base, pending = 0.0, None
        t = ItemTableReplica(1, 1, torch.device("cpu"), rule=rule)
        for _ in range(exchanges):
            for r in range(R):
                for _ in range(n):                     # a slice: n SGD steps on the squared error of the bias alone
                    b[r] += lr * ((target + rs.normal(0, 0.1)) - b[r])
            d = b - base
            if pending is not None:                    # the previous exchange lands one slice late
                Rp, dp = pending
                b += Rp - dp
                base += Rp
                d = b - base
            S = torch.tensor([[d.sum()]], dtype=torch.float32)
            w = torch.tensor([float(R) if rule == "sqrt" else float((d * d).sum())])
            pending = (float((S[:, 0] * t._factors(S, w))[0]), d.copy())
            if not np.isfinite(b).all() or np.abs(b).max() > 1e6:
                return np.inf
        return float(np.abs((base + pending[0]) - target))
    assert run("align") < 0.2    # (hovers around the target within the SGD noise: 2 +- 0.09 over 24 exchanges)
    assert run("sqrt") > 10.0    # (the documented failure of the round-3 rule at this staleness: 5.6, -4.7, 14, -20, 43 ...)


def test_exchanges_per_epoch_rule():
    from cornac_amd.dist import exchanges_per_epoch

    assert exchanges_per_epoch(20_000_263, 26_744) == 16          # the ML-20M shape: the emulated default
    assert exchanges_per_epoch(62_500_000, 10_000_000) == 1       # the configs[4] slice: one overlapped exchange per epoch
    assert exchanges_per_epoch(5_000_000, 26_744) == 4            # a quarter of ML-20M per rank (the emulation's own shape)
    assert exchanges_per_epoch(10 ** 12, 1000) == 64 and exchanges_per_epoch(10, 1000) == 1


def test_single_process_sync_is_a_rebase():
    t = ItemTableReplica(5, 3, torch.device("cpu"))
    t.load(np.ones((5, 3), np.float32), np.zeros(5, np.float32))
    t.V[2] += 4
    t.sync()
    assert torch.equal(t.base, t.flat) and float(t.V[2, 0]) == 5.0


def test_partition_users_by_nnz_and_slice():
    rs = np.random.RandomState(0)
    deg = rs.poisson(20, size=1000)
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    indices = rs.randint(0, 50, size=indptr[-1]).astype(np.int32)
    for world in (1, 2, 4, 8):
        cuts = partition_users_by_nnz(indptr, world)
        assert cuts[0] == 0 and cuts[-1] == 1000 and (np.diff(cuts) > 0).all()
        loads = [indptr[cuts[r + 1]] - indptr[cuts[r]] for r in range(world)]
        assert max(loads) - min(loads) <= 2 * deg.max()
        total = 0
        for r in range(world):
            ip, ix = slice_csr(indptr, indices, cuts[r], cuts[r + 1])
            assert ip[0] == 0 and ip[-1] == len(ix) and ip.dtype == np.int32
            assert np.array_equal(ix, indices[indptr[cuts[r]]:indptr[cuts[r + 1]]])
            total += len(ix)
        assert total == indptr[-1]


# ---- regime 2: row-sharded item table (all-to-all of rows) ----------------------------------------
class _HostRowOps:
    """host stand-in for DeviceRowOps (the HIP gather / scatter-add kernels) on CPU tensors"""

    def gather(self, table, ids, out):
        out.copy_(table[ids.long()])

    def scatter_add(self, table, ids, delta):
        table.index_add_(0, ids.long(), delta)


class _FakeShardTrainer:
    """stands in for _lib.BprTrainer: fixed triplets per rank, `apply` adds recognisable increments"""

    def __init__(self, rank, n_items):
        self.rank, self.n_items, self.batches = rank, n_items, 0

    def make(self, n):
        rs = np.random.RandomState(100 * self.rank + self.batches)
        self.batches += 1
        u = rs.randint(0, 5, n).astype(np.int32)
        u[::7] = -1  # skipped draws
        return u, rs.randint(0, self.n_items, n).astype(np.int32), rs.randint(0, self.n_items, n).astype(np.int32)

    def sync(self):
        return (0, 0)


def _shard_worker(rank, world, port, out):
    from cornac_amd.dist import RowShardedBprTrainer, RowShardedItemTable

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        n_items, k = 11, 3
        V0 = np.arange(n_items, dtype=np.float32)[:, None] * np.ones((1, k), np.float32)
        B0 = -np.arange(n_items, dtype=np.float32)
        # (a) the table alone: fetch returns the owners' rows, push adds every requester's delta
        t = RowShardedItemTable(n_items, k, dev, _HostRowOps())
        t.load(V0, B0)
        want_items = torch.tensor([0, 1, 4, 9, 10] if rank == 0 else [1, 2, 9], dtype=torch.int64)
        uniq = torch.unique(t.owner_major(want_items))
        rows, bias, plan = t.fetch(uniq)
        items_back = (uniq % t.rows_per_rank) * world + uniq // t.rows_per_rank
        assert torch.equal(rows[:, 0], items_back.float()) and torch.equal(bias, -items_back.float())
        t.push(plan, torch.full_like(rows, float(rank + 1)), torch.full_like(bias, 10.0 * (rank + 1)))
        Vf, Bf = t.gather_full()
        # (b) the trainer loop with stand-ins: every valid triplet adds +1 to its i-row, -1 to its j-row, +0.5 bias i
        fake = _FakeShardTrainer(rank, n_items)
        sh = RowShardedBprTrainer(fake, n_items, k, dev, micro_batch=20, ops=_HostRowOps())
        sh.load_items(V0, B0)
        seen = []

        def sample(n):
            u, i, j = fake.make(n)
            seen.append((u.copy(), i.copy(), j.copy()))
            i[u < 0] = -1
            j[u < 0] = -1
            return torch.tensor(u), torch.tensor(i), torch.tensor(j)

        def apply(u, si, sj, rows, bias_pad, lr, reg, use_bias):
            ok = u >= 0            # skipped draws stay in the arrays (u = -1): the apply kernel ignores them
            si, sj = si[ok], sj[ok]
            rows.index_add_(0, si.long(), torch.ones(len(si), k))
            rows.index_add_(0, sj.long(), -torch.ones(len(sj), k))
            bias_pad[:, 0].index_add_(0, si.long(), torch.full((len(si),), 0.5))

        sh._sample, sh._apply = sample, apply
        sh.run(50, 0.1, 0.0)  # micro-batches of 20, 20, 10
        V2, B2 = sh.table.gather_full()
        out[rank] = (Vf.numpy().copy(), Bf.numpy().copy(), V2.numpy().copy(), B2.numpy().copy(), seen, sh.triplets)
    finally:
        dist.destroy_process_group()


def test_row_sharded_table_world2():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_shard_worker, args=(2, port, out), nprocs=2, join=True)
    n_items, k = 11, 3
    V0 = np.arange(n_items, dtype=np.float32)[:, None] * np.ones((1, k), np.float32)
    B0 = -np.arange(n_items, dtype=np.float32)
    Vf0, Bf0, V20, B20, seen0, n0 = out[0]
    Vf1, Bf1, V21, B21, seen1, n1 = out[1]
    assert np.array_equal(Vf0, Vf1) and np.array_equal(V20, V21) and np.array_equal(B20, B21)
    # (a) rows requested by one rank receive its delta, rows requested by both the summed delta / sqrt(2)
    want = V0.copy()
    wb = B0.copy()
    reqs = (([0, 1, 4, 9, 10], 0), ([1, 2, 9], 1))
    senders = np.zeros(n_items)
    for items, r in reqs:
        senders[items] += 1
    for items, r in reqs:
        want[items] += (r + 1) / np.sqrt(senders[items])[:, None]
        wb[items] += 10.0 * (r + 1) / np.sqrt(senders[items])
    assert np.allclose(Vf0, want) and np.allclose(Bf0, wb)
    # (b) the trainer loop: per micro-batch every rank's delta per row, owners apply sum / sqrt(sending ranks)
    want, wb, n_valid = V0.copy(), B0.copy(), 0
    for (u0, i0, j0), (u1, i1, j1) in zip(seen0, seen1):
        dV = [np.zeros_like(V0), np.zeros_like(V0)]
        dB = [np.zeros_like(B0), np.zeros_like(B0)]
        touched = np.zeros(n_items)
        for r, (u, i, j) in enumerate(((u0, i0, j0), (u1, i1, j1))):
            ok = u >= 0
            n_valid += int(ok.sum())
            np.add.at(dV[r], i[ok], 1.0)
            np.add.at(dV[r], j[ok], -1.0)
            np.add.at(dB[r], i[ok], 0.5)
            t = np.zeros(n_items)
            t[np.unique(np.concatenate([i[ok], j[ok]]))] = 1
            touched += t
        div = np.sqrt(np.maximum(touched, 1))
        want += (dV[0] + dV[1]) / div[:, None]
        wb += (dB[0] + dB[1]) / div
    assert [len(b[0]) for b in seen0] == [20, 20, 10] == [len(b[0]) for b in seen1]
    assert n0 + n1 == n_valid
    assert np.allclose(V20, want) and np.allclose(B20, wb)


class _OracleTrainer:
    """stands in for _lib.BprTrainer with real BPR arithmetic: the oracle's sequential epoch function runs `n` draws
    per enqueue IN PLACE on the replica's item table (what the bound device table is on a GPU)"""

    def __init__(self, table, indptr, indices, n_items, k, seed):
        import ctypes as C

        from oracle import oracle as orc

        self.C, self.orc = C, orc
        self.indptr, self.indices = np.ascontiguousarray(indptr, np.int32), np.ascontiguousarray(indices, np.int32)
        self.user_ids = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr)).astype(np.int32)
        self.neg_ids = np.arange(n_items, dtype=np.int32)
        self.V, self.B = table.V.numpy(), table.B.numpy()          # views of the replica's flat buffer
        self.U = ((np.random.RandomState(seed).uniform(0, 1, (len(indptr) - 1, k)).astype(np.float32) - 0.5) / k)
        self.gp, self.gn = orc.MT19937(seed), orc.MT19937(seed + 1)
        self.k, self.n_items, self.correct, self.skipped = k, n_items, 0, 0

    def hogwild_enqueue(self, n, lr, reg, use_bias, neg_population, flags):
        c, s = self.C.c_int64(), self.C.c_int64()
        nnz = len(self.user_ids)
        rc = self.orc.lib().oracle_bpr_epoch_seq(self.gp.ptr, self.gn.ptr, nnz - 1, self.n_items - 1, int(n), self.user_ids,
                                                 self.indices, self.neg_ids, self.indptr, self.U, self.V, self.B, self.k, lr,
                                                 reg, int(use_bias), self.C.byref(c), self.C.byref(s), None, None, None)
        assert rc == 0
        self.correct, self.skipped = self.correct + c.value, self.skipped + s.value

    def sync(self):
        return (self.correct, self.skipped)


def _popularity_data(rank, n_users=250, n_items=90, per_user=24):  # noqa: D401
    """every rank has its own users; all draw from one Zipf item popularity, so the item side is shared knowledge"""
    rs = np.random.RandomState(100 + rank)
    p = 1.0 / np.arange(1, n_items + 1) ** 1.1
    rows = [np.sort(rs.choice(n_items, per_user, replace=False, p=p / p.sum())) for _ in range(n_users)]
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    return indptr, np.concatenate(rows).astype(np.int32), n_items


def _pairwise_accuracy(U, V, B, indptr, indices, n_items, seed=0):
    rs = np.random.RandomState(seed)
    hit = n = 0
    for u in range(len(indptr) - 1):
        pos = indices[indptr[u]:indptr[u + 1]]
        neg = np.setdiff1d(np.arange(n_items), pos)
        i, j = rs.choice(pos, 8), rs.choice(neg, 8)
        s = B + V @ U[u]
        hit, n = hit + int((s[i] > s[j]).sum()), n + 8
    return hit / n


def _learn_worker(rank, world, port, out, sparse_threshold=None, n_items_total=90, per_epoch=8):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        indptr, indices, n_items = _popularity_data(rank, n_items=n_items_total)
        k, nnz = 8, len(indices)
        sh = ShardedBprTrainer(None, total_items=n_items, k=k, device=torch.device("cpu"),
                               sync_every=(nnz + per_epoch - 1) // per_epoch, sparse_threshold=sparse_threshold)
        init = np.random.RandomState(7)                                  # identical item table on every rank
        sh.load_items((init.uniform(0, 1, (n_items, k)).astype(np.float32) - 0.5) / k, np.zeros(n_items, np.float32))
        sh.trainer = _OracleTrainer(sh.table, indptr, indices, n_items, k, seed=11 + rank)
        before = _pairwise_accuracy(sh.trainer.U, sh.table.V.numpy(), sh.table.B.numpy(), indptr, indices, n_items)
        for _ in range(6):
            sh.run(nnz, lr=0.05, reg=0.01)                               # 8 overlapped exchanges per epoch
        correct, skipped = sh.finish()
        after = _pairwise_accuracy(sh.trainer.U, sh.table.V.numpy(), sh.table.B.numpy(), indptr, indices, n_items)
        out[rank] = (sh.table.V.numpy().copy(), sh.table.B.numpy().copy(), before, after, correct, skipped, nnz,
                     sh.table.base.numpy().copy())
        out["exchanges%d" % rank] = dict(sh.table.exchanges)
    finally:
        dist.destroy_process_group()


def test_two_ranks_sparse_exchange_equals_the_dense_one_with_real_bpr_arithmetic():
    """regime 1 with the SPARSE delta exchange (records of the touched rows through all_gather; dense bucket only when a
    rank touched more than the threshold) on two gloo ranks with real BPR arithmetic: a long-tailed catalogue of 2 000
    items and 64 exchanges per epoch, so most exchanges touch a small part of the table.  Same sample streams ->
    the sparse run ends with the same consolidated table as the dense run (to fp32 summation order), and it really went
    through the sparse path."""
    res = {}
    for thr in (None, 0.5):
        out = mp.Manager().dict()
        mp.spawn(_learn_worker, args=(2, _free_port(), out, thr, 2000, 64), nprocs=2, join=True)
        res[thr] = (out[0], out[1], out["exchanges0"])
    (d0, d1, exd), (s0, s1, exs) = res[None], res[0.5]
    assert exd["sparse"] == 0 and exs["sparse"] > 0.9 * (exs["sparse"] + exs["dense"]), (exd, exs)
    assert np.array_equal(s0[7], s1[7]), "both ranks hold the same rebased table"
    # same rule, same per-row arithmetic: only the order in which the ranks' records are summed differs
    assert np.abs(s0[0] - d0[0]).max() < 2e-5 and np.abs(s0[1] - d0[1]).max() < 2e-5, (np.abs(s0[0] - d0[0]).max(), np.abs(s0[1] - d0[1]).max())
    assert s0[3] > 0.7 and abs(s0[3] - d0[3]) < 0.01


def test_two_ranks_learn_one_consolidated_item_table():
    """the whole regime-1 driver (ShardedBprTrainer.run / finish: chunked training, overlapped delta exchange, sqrt
    rule) with REAL BPR arithmetic on two gloo ranks: both end with the same item table, the consolidated model
    ranks each rank's own positives about as well as a single process training on one rank's data alone"""
    out = mp.Manager().dict()
    port = _free_port()
    mp.spawn(_learn_worker, args=(2, port, out), nprocs=2, join=True)
    (V0, B0, before0, after0, c0, s0, nnz0, base0), (V1, B1, before1, after1, _, _, _, base1) = out[0], out[1]
    # one consolidated table after finish(): the rebased copies are bit-identical (same all-reduce result on every rank),
    # the live copies agree to the rounding of  (base + d) + (R - d)  and equal the base (nothing trained since)
    assert np.array_equal(base0, base1)
    assert np.allclose(V0, V1, rtol=0, atol=1e-7) and np.allclose(B0, B1, rtol=0, atol=1e-6)
    assert np.allclose(np.concatenate([V0.ravel(), B0]), base0, rtol=0, atol=1e-6)
    assert np.isfinite(V0).all() and c0 + s0 <= 6 * nnz0 and c0 > 0
    assert before0 < 0.6 and before1 < 0.6 and after0 > 0.75 and after1 > 0.75
    # single process, same work on rank 0's data, no exchange
    indptr, indices, n_items = _popularity_data(0)
    table = ItemTableReplica(n_items, 8, torch.device("cpu"))
    init = np.random.RandomState(7)
    table.load((init.uniform(0, 1, (n_items, 8)).astype(np.float32) - 0.5) / 8, np.zeros(n_items, np.float32))
    solo = _OracleTrainer(table, indptr, indices, n_items, 8, seed=11)
    for _ in range(6):
        solo.hogwild_enqueue(len(indices), 0.05, 0.01, True, 0, 0)
    alone = _pairwise_accuracy(solo.U, table.V.numpy(), table.B.numpy(), indptr, indices, n_items)
    assert after0 > alone - 0.03, (after0, alone)


def _bpr_apply(U, u, si, sj, rows, bias_pad, lr, reg, use_bias):
    """sequential BPR updates (recom_bpr.pyx:240-267) of a micro-batch on the rank's user rows and the STAGED item rows"""
    R, Bp = rows.numpy(), bias_pad.numpy()
    for t in range(len(u)):
        if u[t] < 0:
            continue   # skipped draw
        a, p, q = int(u[t]), int(si[t]), int(sj[t])
        uu, vi, vj = U[a].copy(), R[p].copy(), R[q].copy()
        z = 1.0 / (1.0 + np.exp(Bp[p, 0] - Bp[q, 0] + uu @ (vi - vj)))
        U[a] += lr * (z * (vi - vj) - reg * uu)
        R[p] += lr * (z * uu - reg * vi)
        R[q] += lr * (-z * uu - reg * vj)
        if use_bias:
            Bp[p, 0] += lr * (z - reg * Bp[p, 0])
            Bp[q, 0] += lr * (-z - reg * Bp[q, 0])


def _sharded_learn_worker(rank, world, port, out, pipeline=False):
    from cornac_amd.dist import RowShardedBprTrainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        indptr, indices, n_items = _popularity_data(rank)
        k, nnz = 8, len(indices)
        user_ids = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
        rs = np.random.RandomState(50 + rank)
        U = ((np.random.RandomState(11 + rank).uniform(0, 1, (len(indptr) - 1, k)).astype(np.float32) - 0.5) / k)
        positives = [set(indices[indptr[a]:indptr[a + 1]].tolist()) for a in range(len(indptr) - 1)]
        sh = RowShardedBprTrainer(_FakeShardTrainer(rank, n_items), n_items, k, torch.device("cpu"), micro_batch=1500,
                                  ops=_HostRowOps(), pipeline=pipeline)
        assert (sh.group_b is not sh.group) == pipeline   # the push side exchanges through its own communicator
        init = np.random.RandomState(7)
        sh.load_items((init.uniform(0, 1, (n_items, k)).astype(np.float32) - 0.5) / k, np.zeros(n_items, np.float32))

        def sample(n):            # uniform positive interaction, uniform negative item, -1 where the draw is a positive
            ii = rs.randint(0, nnz, n)
            u, i, j = user_ids[ii].astype(np.int32), indices[ii].astype(np.int32), rs.randint(0, n_items, n).astype(np.int32)
            skip = np.array([int(b) in positives[a] for a, b in zip(u, j)])
            u, i, j = u.copy(), i.copy(), j.copy()
            u[skip] = i[skip] = j[skip] = -1
            return torch.tensor(u), torch.tensor(i), torch.tensor(j)

        sh._sample = sample
        sh._apply = lambda u, si, sj, rows, bias_pad, lr, reg, use_bias: _bpr_apply(U, u.numpy(), si.numpy(), sj.numpy(), rows,
                                                                                  bias_pad, lr, reg, use_bias)
        V, B = sh.table.gather_full()
        before = _pairwise_accuracy(U, V.numpy(), B.numpy(), indptr, indices, n_items)
        for _ in range(6):
            sh.run(nnz, 0.05, 0.01)       # 4 micro-batches per epoch: fetch -> local updates on staged rows -> push deltas
        V, B = sh.table.gather_full()
        out[rank] = (V.numpy().copy(), B.numpy().copy(), before,
                     _pairwise_accuracy(U, V.numpy(), B.numpy(), indptr, indices, n_items), sh.rows_fetched, sh.triplets)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pipeline", [False, True])
def test_two_ranks_learn_over_a_row_sharded_item_table(pipeline):
    """regime 2 end to end on two gloo ranks with real BPR arithmetic on the staged rows: items live on their owner
    rank, every micro-batch fetches the touched rows, updates them locally and returns them (the owner applies the
    difference); the assembled table is the same on both ranks and ranks each rank's own positives well.  pipeline=True
    is the GPU driver's order (update / push of micro-batch r-1 enqueued before the fetch of r, second communicator)"""
    out = mp.Manager().dict()
    mp.spawn(_sharded_learn_worker, args=(2, _free_port(), out, pipeline), nprocs=2, join=True)
    (V0, B0, before0, after0, fetched0, n0), (V1, B1, before1, after1, _, _) = out[0], out[1]
    assert np.array_equal(V0, V1) and np.array_equal(B0, B1) and np.isfinite(V0).all()
    assert before0 < 0.6 and before1 < 0.6 and after0 > 0.75 and after1 > 0.75
    assert 0 < fetched0 <= 24 * 90 and n0 > 0          # de-duplicated requests: at most every item once per micro-batch


def _uneven_worker(rank, world, port, out, pipeline=False):
    from cornac_amd.dist import RowShardedBprTrainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_items, k = 40, 4
        sh = RowShardedBprTrainer(_FakeShardTrainer(rank, n_items), n_items, k, torch.device("cpu"), micro_batch=100,
                                  ops=_HostRowOps(), pipeline=pipeline)
        sh.load_items(np.zeros((n_items, k), np.float32), np.zeros(n_items, np.float32))
        rs = np.random.RandomState(rank)
        calls = []

        def sample(n):
            calls.append(n)
            return (torch.tensor(rs.randint(0, 5, n).astype(np.int32)), torch.tensor(rs.randint(0, n_items, n).astype(np.int32)),
                    torch.tensor(rs.randint(0, n_items, n).astype(np.int32)))

        def apply(u, si, sj, rows, bias_pad, lr, reg, use_bias):
            rows.index_add_(0, si.long(), torch.ones(len(si), rows.shape[1]))    # every positive row +1, every negative
            rows.index_add_(0, sj.long(), -torch.ones(len(sj), rows.shape[1]))   # row -1: the table's total stays 0

        sh._sample, sh._apply = sample, apply
        sh.run(350 if rank == 0 else 120, 0.05, 0.01)   # 4 micro-batches on rank 0, 2 on rank 1
        V, _ = sh.table.gather_full()
        out[rank] = (calls, float(V.sum()), float(V.abs().sum()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pipeline", [False, True])
def test_row_sharded_ranks_with_different_draw_counts_do_not_deadlock(pipeline):
    """every micro-batch is a collective; ranks whose user shards hold different numbers of interactions
    agree on the round count up front and the rank that runs dry serves empty rounds"""
    out = mp.Manager().dict()
    mp.spawn(_uneven_worker, args=(2, _free_port(), out, pipeline), nprocs=2, join=True)
    (calls0, tot0, mass0), (calls1, tot1, mass1) = out[0], out[1]
    assert calls0 == [100, 100, 100, 50] and calls1 == [100, 20]
    assert tot0 == tot1 and mass0 == mass1 and mass0 > 0


def _rank_worker(rank, world, port, out):
    from cornac_amd.dist import rank_users_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(3)                      # the same (replicated) tables on every rank
        U, V = rs.normal(size=(37, 6)).astype(np.float32), rs.normal(size=(50, 6)).astype(np.float32)
        seen = []

        def rank_fn(users):
            seen.append(np.asarray(users).copy())
            S = U[users] @ V.T
            items = np.argsort(-S, axis=1, kind="stable")[:, :5].astype(np.int32)
            return items, np.take_along_axis(S, items, 1).astype(np.float32)

        users = np.arange(37)[::-1].copy()
        items, scores = rank_users_sharded(rank_fn, users, 5)
        out[rank] = (items, scores, np.concatenate(seen) if seen else np.empty(0, np.int64))
    finally:
        dist.destroy_process_group()


def test_user_block_sharded_ranking_gathers_the_topk_lists():
    """scoring shards by user block, then one gather of the top-k lists — every rank ends with the
    ranking a single process computes, each having scored only its own block"""
    out = mp.Manager().dict()
    mp.spawn(_rank_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    (i0, s0, seen0), (i1, s1, seen1) = out[0], out[1]
    rs = np.random.RandomState(3)
    U, V = rs.normal(size=(37, 6)).astype(np.float32), rs.normal(size=(50, 6)).astype(np.float32)
    users = np.arange(37)[::-1]
    S = U[users] @ V.T
    want = np.argsort(-S, axis=1, kind="stable")[:, :5]
    assert np.array_equal(i0, want) and np.array_equal(i1, want) and np.allclose(s0, np.take_along_axis(S, want, 1))
    assert len(seen0) + len(seen1) == 37 and set(seen0.tolist()).isdisjoint(seen1.tolist())


# ---- MF over the replicated item side (ShardedMfTrainer) ------------------------------------------------------------
class _OracleMfTrainer:
    """stands in for _lib.MfTrainer with real MF arithmetic: the oracle's sequential fit_sgd loop (backend_cpu.pyx:56-90)
    runs the ratings of a slice IN PLACE on the replica's item side"""

    def __init__(self, table, rid, cid, val, n_users, k, seed):
        from oracle import oracle as orc

        self.orc = orc
        self.rid, self.cid = np.ascontiguousarray(rid, np.int64), np.ascontiguousarray(cid, np.int64)
        self.val = np.ascontiguousarray(val, np.float32)
        self.V, self.Bi = table.V.numpy(), table.B.numpy()
        rs = np.random.RandomState(seed)
        self.U = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
        self.Bu = np.zeros(n_users, np.float32)
        self.k, self.sq = k, 0.0

    def epoch_enqueue(self, part, n_parts, lr, reg, mu, use_bias=True):
        nnz = len(self.val)
        s0, s1 = nnz * part // n_parts, nnz * (part + 1) // n_parts
        if s1 == s0:
            return
        loss = np.zeros(1, np.float32)
        self.orc.lib().oracle_mf_fit(self.rid[s0:s1].copy(), self.cid[s0:s1].copy(), self.val[s0:s1].copy(), s1 - s0, self.U,
                                     self.V, self.Bu, self.Bi, self.k, lr, reg, mu, 1, 1, int(use_bias), 0, loss.ctypes.data)
        self.sq += 2.0 * float(loss[0])

    def sync(self):
        sq, self.sq = self.sq, 0.0
        return sq


def _mf_data(rank, n_users=200, n_items=70, per_user=20):
    """every rank has its own users; ratings = a shared low-rank item structure + user taste + noise"""
    rs_items = np.random.RandomState(5)
    q = rs_items.normal(0, 1, (n_items, 3))
    item_bias = rs_items.normal(0, 0.5, n_items)
    rs = np.random.RandomState(200 + rank)
    p = 1.0 / np.arange(1, n_items + 1) ** 0.9
    rid, cid, val = [], [], []
    for u in range(n_users):
        items = np.sort(rs.choice(n_items, per_user, replace=False, p=p / p.sum()))
        taste = rs.normal(0, 1, 3)
        r = 3.0 + item_bias[items] + 0.6 * q[items] @ taste + rs.normal(0, 0.3, per_user)
        rid += [u] * per_user
        cid += list(items)
        val += list(np.clip(r, 1, 5))
    return np.array(rid, np.int64), np.array(cid, np.int64), np.array(val, np.float32), n_users, n_items


def _mf_rmse(U, V, Bu, Bi, mu, rid, cid, val):
    pred = mu + Bu[rid] + Bi[cid] + np.einsum("nk,nk->n", U[rid], V[cid])
    return float(np.sqrt(np.mean((pred - val) ** 2)))


def _mf_worker(rank, world, port, out, sparse_threshold=None):
    from cornac_amd.dist import ShardedMfTrainer, global_mean_across_ranks

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rid, cid, val, n_users, n_items = _mf_data(rank)
        k = 6
        mu = global_mean_across_ranks(val)
        sh = ShardedMfTrainer(None, total_items=n_items, k=k, device=torch.device("cpu"), parts_per_epoch=8,
                              sparse_threshold=sparse_threshold)
        init = np.random.RandomState(7)
        sh.load_items(init.normal(0, 0.01, (n_items, k)).astype(np.float32), np.zeros(n_items, np.float32))
        sh.trainer = _OracleMfTrainer(sh.table, rid, cid, val, n_users, k, seed=31 + rank)
        tr = sh.trainer
        before = _mf_rmse(tr.U, tr.V, tr.Bu, tr.Bi, mu, rid, cid, val)
        losses = []
        for _ in range(25):
            sh.run_epoch(lr=0.02, reg=0.02, mu=mu)
            losses.append(sh.finish())
        after = _mf_rmse(tr.U, tr.V, tr.Bu, tr.Bi, mu, rid, cid, val)
        out[rank] = (sh.table.V.numpy().copy(), sh.table.B.numpy().copy(), sh.table.base.numpy().copy(), before, after, mu,
                     losses, dict(sh.table.exchanges))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sparse_threshold", [None, 1.0])
def test_two_ranks_train_mf_over_one_replicated_item_side(sparse_threshold):
    """ShardedMfTrainer on two gloo ranks with REAL MF arithmetic (the oracle's fit_sgd loop on every slice): users split
    across the ranks, [V | Bi] reconciled 8 times per epoch (dense all-reduce, and the sparse record exchange).  Both
    ranks end with the same item side, the global mean is the mean of the union, the per-epoch loss falls, and each
    rank's own ratings are fitted about as well as by a single process training on that rank's data alone."""
    out = mp.Manager().dict()
    mp.spawn(_mf_worker, args=(2, _free_port(), out, sparse_threshold), nprocs=2, join=True)
    (V0, B0, base0, before0, after0, mu0, loss0, ex0), (V1, B1, base1, before1, after1, mu1, loss1, _) = out[0], out[1]
    assert np.array_equal(base0, base1)
    assert np.allclose(V0, V1, rtol=0, atol=1e-6) and np.allclose(B0, B1, rtol=0, atol=1e-6)
    all_val = np.concatenate([_mf_data(r)[2] for r in (0, 1)])
    assert mu0 == mu1 and abs(mu0 - float(all_val.astype(np.float64).mean())) < 1e-9
    assert (ex0["sparse"] > 0) == (sparse_threshold is not None) and ex0["dense"] + ex0["sparse"] >= 25 * 8
    assert loss0[-1] < 0.5 * loss0[0] and loss1[-1] < 0.5 * loss1[0]
    assert before0 > 0.8 and after0 < 0.6 * before0 and after1 < 0.6 * before1, (before0, after0, after1)
    # single process on rank 0's data, same slices, no exchange
    rid, cid, val, n_users, n_items = _mf_data(0)
    table = ItemTableReplica(n_items, 6, torch.device("cpu"))
    init = np.random.RandomState(7)
    table.load(init.normal(0, 0.01, (n_items, 6)).astype(np.float32), np.zeros(n_items, np.float32))
    solo = _OracleMfTrainer(table, rid, cid, val, n_users, 6, seed=31)
    for _ in range(25):
        for part in range(8):
            solo.epoch_enqueue(part, 8, 0.02, 0.02, mu0)
    alone = _mf_rmse(solo.U, solo.V, solo.Bu, solo.Bi, mu0, rid, cid, val)
    assert after0 < alone + 0.05, (after0, alone)


def _mf_uneven_worker(rank, world, port, out):
    from cornac_amd.dist import ShardedMfTrainer, global_mean_across_ranks

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 1 holds a tenth of rank 0's users, and with the default slicing (8 exchanges per epoch up to 4 ranks) its
        # slices are a few dozen ratings — some exchanges touch few rows (sparse records), some many (dense bucket)
        rid, cid, val, n_users, n_items = _mf_data(rank, n_users=200 if rank == 0 else 20)
        mu = global_mean_across_ranks(val)
        sh = ShardedMfTrainer(None, total_items=n_items, k=5, device=torch.device("cpu"), sparse_threshold=0.5)
        assert sh.parts == 8 and sh.table.rule == "align"
        init = np.random.RandomState(7)
        sh.load_items(init.normal(0, 0.01, (n_items, 5)).astype(np.float32), np.zeros(n_items, np.float32))
        sh.trainer = _OracleMfTrainer(sh.table, rid, cid, val, n_users, 5, seed=31 + rank)
        for _ in range(6):
            sh.run_epoch(lr=0.02, reg=0.02, mu=mu)
            sh.finish()
        tr = sh.trainer
        out[rank] = (sh.table.base.numpy().copy(), _mf_rmse(tr.U, tr.V, tr.Bu, tr.Bi, mu, rid, cid, val), dict(sh.table.exchanges))
    finally:
        dist.destroy_process_group()


def test_mf_ranks_of_very_different_sizes_exchange_in_lockstep():
    """the number of exchanges per epoch is a property of the driver (parts_per_epoch), not of a rank's data: a rank with a
    tenth of the ratings runs the same 8 collectives per epoch (no deadlock, no mismatch), the dense / sparse decision is
    taken collectively, and both ranks end on one item side"""
    out = mp.Manager().dict()
    mp.spawn(_mf_uneven_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    (base0, rmse0, ex0), (base1, rmse1, ex1) = out[0], out[1]
    assert np.array_equal(base0, base1) and np.isfinite(base0).all()
    assert ex0["dense"] == ex1["dense"] and ex0["sparse"] == ex1["sparse"] and ex0["dense"] + ex0["sparse"] >= 6 * 8
    assert rmse0 < 1.0 and rmse1 < 1.2


# ---- model-level entry points: fit_bpr_sharded / fit_mf_sharded --------------------------------------------------------
class _BprRankTrainer(_OracleTrainer):
    """the trainer surface fit_bpr_sharded drives, over the oracle's sequential BPR loop on the replica's item table"""

    def __init__(self, table, indptr, indices, n_local, n_items, total_items, k):
        super().__init__(table, indptr, indices, n_items, k, seed=1)

    def set_factors(self, U, V, B):
        self.U = np.array(U, np.float32)

    def seed_hogwild(self, seed):
        self.gp, self.gn = self.orc.MT19937(seed % (2 ** 31)), self.orc.MT19937((seed >> 32) % (2 ** 31) + 1)

    def get_user_factors(self):
        return self.U.copy()

    def close(self):
        pass


class _MfRankTrainer(_OracleMfTrainer):
    def __init__(self, table, rid, cid, val, n_local, n_items, k):
        super().__init__(table, rid, cid, val, n_local, k, seed=1)

    def set_factors(self, U, V, Bu, Bi):
        self.U, self.Bu = np.array(U, np.float32), np.array(Bu, np.float32)

    def get_factors(self):
        return self.U.copy(), None, self.Bu.copy(), None

    def close(self):
        pass


def _model_data(seed=0, nu=160, ni=60, per_user=18):
    rs = np.random.RandomState(seed)
    p = 1.0 / np.arange(1, ni + 1) ** 0.9
    q, ib = rs.normal(0, 1, (ni, 3)), rs.normal(0, 0.5, ni)
    rows = []
    for u in range(nu):
        t = rs.normal(0, 1, 3)
        for i in rs.choice(ni, per_user + (u % 7), replace=False, p=p / p.sum()):
            rows.append(("u%d" % u, "i%d" % i, float(np.clip(np.rint(3 + ib[i] + 0.6 * q[i] @ t + rs.normal(0, 0.3)), 1, 5))))
    return rows


def _fit_sharded_worker(rank, world, port, out):
    import cornac_amd as ca
    from cornac_amd.dist import fit_bpr_sharded, fit_mf_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = ca.Dataset.from_uir(_model_data(), seed=3)
        # the ranks' own generators differ on purpose (seed = rank): rank 0's initial tables must win
        bpr = ca.BPR(k=6, max_iter=8, learning_rate=0.05, lambda_reg=0.01, seed=rank, mode="hogwild")
        fit_bpr_sharded(bpr, ds, sync_per_epoch=8, trainer_factory=_BprRankTrainer)
        mf = ca.MF(k=6, max_iter=15, learning_rate=0.02, lambda_reg=0.02, seed=rank, mode="hogwild")
        fit_mf_sharded(mf, ds, parts_per_epoch=8, trainer_factory=_MfRankTrainer)
        out[rank] = dict(bU=bpr.u_factors.copy(), bV=bpr.i_factors.copy(), bB=bpr.i_biases.copy(), bstats=bpr.fit_stats,
                         mU=mf.u_factors.copy(), mV=mf.i_factors.copy(), mBu=mf.u_biases.copy(), mBi=mf.i_biases.copy(),
                         mloss=mf.loss_history.copy(), mu=float(mf.global_mean))
        with pytest.raises(ValueError):
            fit_bpr_sharded(ca.BPR(k=4, seed=1), ds, trainer_factory=_BprRankTrainer)   # seeded => sequential semantics
    finally:
        dist.destroy_process_group()


def test_model_level_sharded_fits_return_one_complete_model_on_every_rank():
    """fit_bpr_sharded / fit_mf_sharded on two gloo ranks with real arithmetic behind the trainer surface: the users are
    cut by interaction count, every rank trains its range, and BOTH ranks return with the same complete model — user rows
    of both ranges gathered, one item side — that has learnt (pairwise accuracy / RMSE on all the data), starting from
    rank 0's initial tables although the ranks' generators differ."""
    import cornac_amd as ca

    out = mp.Manager().dict()
    mp.spawn(_fit_sharded_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = out[0], out[1]
    for name in ("bU", "bV", "bB", "mU", "mV", "mBu", "mBi"):
        assert np.allclose(a[name], b[name], rtol=0, atol=1e-6), name
    assert a["bstats"] == b["bstats"] and a["bstats"][0][0] > 0
    assert np.allclose(a["mloss"], b["mloss"]) and a["mloss"][-1] < 0.85 * a["mloss"][0] and np.all(np.diff(a["mloss"]) < 0)
    ds = ca.Dataset.from_uir(_model_data(), seed=3)
    X = ds.matrix
    # BPR: positives outrank random non-positives for users of BOTH ranges
    rs, hit, n = np.random.RandomState(0), 0, 0
    for u in range(ds.num_users):
        pos = X.indices[X.indptr[u]:X.indptr[u + 1]]
        neg = np.setdiff1d(np.arange(ds.num_items), pos)
        s = a["bB"] + a["bV"] @ a["bU"][u]
        i, j = rs.choice(pos, 6), rs.choice(neg, 6)
        hit, n = hit + int((s[i] > s[j]).sum()), n + 6
    assert hit / n > 0.68, hit / n   # (8 epochs from a cold start; 0.5 = chance)
    rid, cid, val = ds.uir_tuple
    pred = a["mu"] + a["mBu"][rid] + a["mBi"][cid] + np.einsum("nk,nk->n", a["mU"][rid], a["mV"][cid])
    rmse, rmse_mean_only = float(np.sqrt(np.mean((pred - val) ** 2))), float(np.sqrt(np.mean((a["mu"] - val) ** 2)))
    assert rmse < 0.9 * rmse_mean_only, (rmse, rmse_mean_only)
    for lo, hi in ((0, 20), (ds.num_users - 20, ds.num_users)):     # users of both ranks' ranges were trained and gathered
        assert np.abs(a["mBu"][lo:hi]).max() > 1e-3 and np.abs(a["bU"][lo:hi]).max() > 0.02


# ---- resident exchange (ShardedBprTrainer.run_epoch, the protocol of csrc/bpr_ldsbin.inc's EXCH kernels) -------------
def _resident_worker(rank, world, port, out, resident, lag, epochs, parts):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        indptr, indices, n_items = _popularity_data(rank)
        k, nnz = 8, len(indices)
        sh = ShardedBprTrainer(None, total_items=n_items, k=k, device=torch.device("cpu"), sync_every=nnz)
        sh.resident_lag = lag
        init = np.random.RandomState(7)
        sh.load_items((init.uniform(0, 1, (n_items, k)).astype(np.float32) - 0.5) / k, np.zeros(n_items, np.float32))
        sh.trainer = _OracleTrainer(sh.table, indptr, indices, n_items, k, seed=11 + rank)
        for _ in range(epochs):
            sh.run_epoch(nnz, parts, 0.05, 0.01, resident=resident)
        sh.finish()
        acc = _pairwise_accuracy(sh.trainer.U, sh.table.V.numpy(), sh.table.B.numpy(), indptr, indices, n_items)
        out[rank] = (sh.table.V.numpy().copy(), sh.table.B.numpy().copy(), sh.table.base.numpy().copy(), acc,
                     dict(sh.table.exchanges))
    finally:
        dist.destroy_process_group()


def _resident_run(resident, lag=1, epochs=1, parts=8):
    out = mp.Manager().dict()
    mp.spawn(_resident_worker, args=(2, _free_port(), out, resident, lag, epochs, parts), nprocs=2, join=True)
    return out[0], out[1]


def test_resident_exchange_equals_the_chunk_protocol_within_an_epoch():
    """The resident protocol (publish d = flat - base, base = flat; apply c = rule(S) - d_own to flat AND base whenever the
    sum has landed) with every exchange applied one boundary after its publication is the overlapped chunk protocol
    (begin_sync / finish_sync) in other words: two gloo ranks, real BPR arithmetic, one epoch of 8 exchanges -> the same
    consolidated table to fp32 rounding, the same number of collectives, replicas rebased."""
    (Va, Ba, base_a, _, ex_a), (Va1, Ba1, base_a1, _, _) = _resident_run(True)
    (Vb, Bb, base_b, _, ex_b), _ = _resident_run(False)
    assert ex_a["dense"] == ex_b["dense"] == 8 and ex_a["resident"] == 8, (ex_a, ex_b)
    assert np.abs(Va - Vb).max() < 2e-6 and np.abs(Ba - Bb).max() < 2e-6, (np.abs(Va - Vb).max(), np.abs(Ba - Bb).max())
    # one consolidated table on both ranks; nothing unpublished is left after the epoch's flush
    assert np.allclose(Va, Va1, rtol=0, atol=1e-6) and np.allclose(Ba, Ba1, rtol=0, atol=1e-6)
    assert np.array_equal(np.concatenate([Va.ravel(), Ba]), base_a) and np.allclose(base_a, base_a1, rtol=0, atol=1e-6)


@pytest.mark.parametrize("lag", [1, 3])
def test_resident_exchange_learns_with_late_landing_sums(lag):
    """several exchanges in flight (a sum applied three boundaries after its publication, the rest by the epoch's flush):
    the accounting stays exact — both ranks end with one table — and the consolidated model is as good as the chunk
    protocol's"""
    (V0, B0, base0, acc0, _), (V1, B1, _, acc1, _) = _resident_run(True, lag=lag, epochs=6)
    (_, _, _, ref0, _), (_, _, _, ref1, _) = _resident_run(False, epochs=6)
    assert np.isfinite(V0).all() and np.allclose(V0, V1, rtol=0, atol=1e-5) and np.allclose(B0, B1, rtol=0, atol=1e-5)
    assert np.array_equal(np.concatenate([V0.ravel(), B0]), base0)
    assert acc0 > 0.75 and acc1 > 0.75 and acc0 > ref0 - 0.03 and acc1 > ref1 - 0.03, (acc0, acc1, ref0, ref1)


def test_resident_exchange_of_one_process_keeps_its_own_steps_bit_for_bit():
    """no process group: S = d and one touching rank -> c = 1 * d - d = 0 exactly; the table evolves as if nothing were
    exchanged; with a twin rank played by the bucket hook (S = 2 d, two touching ranks) every published delta is
    amplified to sqrt(2) d — checked in closed form on a stand-in with known steps"""
    dev = torch.device("cpu")
    sh = ShardedBprTrainer(None, total_items=6, k=4, device=dev, sync_every=100)
    sh.trainer = _FakeTrainer(sh.table, 0)
    V0 = np.arange(24, dtype=np.float32).reshape(6, 4)
    sh.load_items(V0, np.zeros(6, np.float32))
    sh.run_epoch(300, 3, lr=0.01, reg=0.0, resident=True)
    sh.finish()
    want = V0.copy()
    want[0::2] += 0.01 * 300
    want[0] += 3.0
    assert np.array_equal(sh.table.V.numpy(), want) and np.array_equal(sh.table.B.numpy(), np.full(6, 1.5, np.float32))
    assert torch.equal(sh.table.flat, sh.table.base)
    # twin rank: every bucket doubled, weights doubled
    sh2 = ShardedBprTrainer(None, total_items=6, k=4, device=dev, sync_every=100)
    sh2.trainer = _FakeTrainer(sh2.table, 0)
    sh2.load_items(V0, np.zeros(6, np.float32))
    sh2.resident_bucket_hook = lambda e, bucket: bucket.mul_(2.0)
    sh2.run_epoch(300, 3, lr=0.01, reg=0.0, resident=True)
    sh2.finish()
    r2 = np.float32(np.sqrt(2.0))
    want2 = V0.copy()
    want2[0::2] += r2 * np.float32(0.01 * 300)
    want2[0] += r2 * 3.0
    assert np.allclose(sh2.table.V.numpy(), want2, rtol=0, atol=1e-5)
    assert np.allclose(sh2.table.B.numpy(), r2 * 1.5, atol=1e-6) and torch.equal(sh2.table.flat, sh2.table.base)


def test_exchange_schedule_rule():
    """dense item sides: several exchanges per epoch under "sqrt" from 16 on, "align" below; sparse item sides (the
    configs[4] slice): one exchange per epoch under "align" — and regime "auto" takes the conveyor there"""
    from cornac_amd.dist import exchange_schedule, exchanges_per_epoch, prefers_conveyor

    assert exchange_schedule(20_000_263, 26_744) == (16, 1, "sqrt") and exchanges_per_epoch(20_000_263, 26_744) == 16
    assert exchange_schedule(5_000_000, 26_744) == (4, 1, "align")
    assert exchange_schedule(62_500_000, 10_000_000) == (1, 1, "align")       # (round 6: no multi-epoch interval by default —
    assert exchange_schedule(62_500_000, 10_000_000, max_epochs=4) == (1, 4, "align")   # the device measurement overturned it)
    assert prefers_conveyor(62_500_000, 10_000_000) and not prefers_conveyor(20_000_263 // 8, 26_744)
    assert exchange_schedule(62_500_000, 2_000_000) == (1, 1, "align")        # 62.5 updates per row and epoch: every epoch
    assert exchange_schedule(10_000_000, 10_000_000, max_epochs=8) == (1, 8, "align")
    assert exchange_schedule(4_000_000_000, 26_744)[0] == 64


def _interval_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        indptr, indices, n_items = _popularity_data(rank)
        k, nnz = 8, len(indices)
        sh = ShardedBprTrainer(None, total_items=n_items, k=k, device=torch.device("cpu"), sync_every=nnz, rule="align")
        init = np.random.RandomState(7)
        sh.load_items((init.uniform(0, 1, (n_items, k)).astype(np.float32) - 0.5) / k, np.zeros(n_items, np.float32))
        sh.trainer = _OracleTrainer(sh.table, indptr, indices, n_items, k, seed=11 + rank)
        for _ in range(5):                                         # exchanges after epochs 2 and 4, the fifth epoch's by finish()
            sh.run_epoch(nnz, 1, 0.05, 0.01, epochs_per_exchange=2)
        sh.finish()
        acc = _pairwise_accuracy(sh.trainer.U, sh.table.V.numpy(), sh.table.B.numpy(), indptr, indices, n_items)
        out[rank] = (sh.table.V.numpy().copy(), sh.table.base.numpy().copy(), acc, dict(sh.table.exchanges))
    finally:
        dist.destroy_process_group()


def test_multi_epoch_exchange_interval_world2():
    """sparse schedule: one exchange every 2 epochs under the align rule, two gloo ranks with real BPR arithmetic — 3
    exchanges for 5 epochs (the last, partial interval is exchanged by finish()), one consolidated table, a model that
    has learnt"""
    out = mp.Manager().dict()
    mp.spawn(_interval_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    (V0, base0, acc0, ex0), (V1, base1, acc1, _) = out[0], out[1]
    assert ex0["dense"] == 3, ex0
    assert np.array_equal(base0, base1) and np.allclose(V0, V1, rtol=0, atol=1e-6)
    assert acc0 > 0.75 and acc1 > 0.75, (acc0, acc1)


# ---- regime 2: the ring conveyor, its blocks bin ranges of the epoch's deal (BinConveyorBprTrainer) ------------------------
class _ConveyorHostTrainer:
    """host stand-in of a rank's conveyor handle (cornac_hip_bpr_conveyor_*): the layout is the oracle's restatement of the
    device's deal; what a launch does to the rows is left to the subclasses"""

    def __init__(self, indptr, indices, n_users, n_items, k, U, cap_target=7):
        from oracle import oracle as orc

        self.orc = orc
        self.indptr, self.indices = np.ascontiguousarray(indptr, np.int32), np.ascontiguousarray(indices, np.int32)
        self.user_ids = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr)).astype(np.int32)
        self.n_items, self.k, self.U, self.cap_target = int(n_items), int(k), U.numpy(), cap_target
        self.correct = self.skipped = 0
        self._layouts = {}

    def seed_hogwild(self, seed):
        self.gp, self.gn = self.orc.MT19937(seed % (2 ** 31)), self.orc.MT19937((seed >> 32) % (2 ** 31) + 1)

    def conveyor_setup(self, n_blocks, rank_item, deal_seed):
        per_block = max(1, -(-self.n_items // (self.cap_target * n_blocks)))
        self.n_bins = n_blocks * per_block
        self.bpb, self.cap = per_block, -(-self.n_items // self.n_bins)
        self.deal_seed = deal_seed
        self.rank_item = (np.argsort(-np.bincount(self.indices, minlength=self.n_items), kind="stable") if rank_item is None
                          else np.asarray(rank_item)).astype(np.int32)
        return self.n_bins, self.bpb, self.cap

    def _layout(self, layout_epoch):
        if layout_epoch not in self._layouts:
            self._layouts[layout_epoch] = self.orc.ldsbin_layout(self.deal_seed, layout_epoch, self.n_bins, self.n_items, self.rank_item)
        return self._layouts[layout_epoch]

    def conveyor_layout(self, layout_epoch):
        si, isl = self._layout(layout_epoch)
        return torch.as_tensor(si.copy()), torch.as_tensor(isl.copy())

    def conveyor_enqueue(self, epoch, layout_epoch, blocks, bufs, lr, reg, use_bias, neg_population, flags):
        W = self.bpb * self.cap
        slot_item, item_slot = self._layout(layout_epoch)
        for blk, buf in zip(blocks, bufs):
            flat = buf.numpy()
            self.train_block(blk, slot_item[blk * W: (blk + 1) * W], item_slot, flat[: W * self.k].reshape(W, self.k), flat[W * self.k:],
                             lr, reg, use_bias)

    def sync(self):
        out, self.correct, self.skipped = (self.correct, self.skipped), 0, 0
        return out

    def close(self):
        pass


class _ConveyorMarkTrainer(_ConveyorHostTrainer):
    """adds a rank-dependent constant to every row of the blocks it is handed: the conveyor's bookkeeping made visible"""

    def __init__(self, rank, log, *a):
        super().__init__(*a)
        self.rank, self.log = rank, log

    def train_block(self, blk, items, item_slot, V, B, lr, reg, use_bias):
        ok = items >= 0
        V[ok] += float(self.rank + 1)
        B[ok] += 10.0 * (self.rank + 1)
        self.correct += int(ok.sum())
        self.log.append(blk)


class _ConveyorOracleTrainer(_ConveyorHostTrainer):
    """real BPR arithmetic on one block: the oracle's sequential epoch over the rank's interactions with the block's items
    (ids = the rows' slots in the block's buffer, negatives among the block's items), in place on the shared user table and
    on the block's buffer"""

    def train_block(self, blk, items, item_slot, V, B, lr, reg, use_bias):
        import ctypes as C

        W = len(items)
        slots = item_slot[self.indices].astype(np.int64)
        mine = (slots // W) == blk
        if not mine.any():
            return
        users, local = self.user_ids[mine], (slots[mine] - blk * W).astype(np.int32)
        order = np.lexsort((local, users))
        users, local = np.ascontiguousarray(users[order]), np.ascontiguousarray(local[order])
        indptr = np.concatenate([[0], np.cumsum(np.bincount(users, minlength=len(self.indptr) - 1))]).astype(np.int32)
        neg_ids = np.flatnonzero(items >= 0).astype(np.int32)
        c, s = C.c_int64(), C.c_int64()
        rc = self.orc.lib().oracle_bpr_epoch_seq(self.gp.ptr, self.gn.ptr, len(users) - 1, len(neg_ids) - 1, len(users), users, local,
                                                 neg_ids, indptr, self.U, V, B, self.k, lr, reg, int(use_bias), C.byref(c),
                                                 C.byref(s), None, None, None)
        assert rc == 0
        self.correct, self.skipped = self.correct + c.value, self.skipped + s.value


def _ring_data(rank, n_users=120, n_items=50, per_user=14):
    rs = np.random.RandomState(300 + rank)
    p = 1.0 / np.arange(1, n_items + 1) ** 0.8
    rows = [np.sort(rs.choice(n_items, per_user, replace=False, p=p / p.sum())) for _ in range(n_users)]
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    return indptr, np.concatenate(rows).astype(np.int32)


def _ring_tables(n_users, n_items, k, rank):
    rs = np.random.RandomState(7)
    V0 = ((rs.uniform(0, 1, (n_items, k)) - 0.5) / k).astype(np.float32)
    B0 = rs.normal(0, 0.01, n_items).astype(np.float32)
    U0 = ((np.random.RandomState(70 + rank).uniform(0, 1, (n_users, k)) - 0.5) / k).astype(np.float32)
    return U0, V0, B0


def _ring_order(world, n_items):
    """the popularity order of the whole matrix (every rank computes the same one)"""
    deg = sum(np.bincount(_ring_data(r, n_items=n_items)[1], minlength=n_items) for r in range(world))
    return np.argsort(-deg, kind="stable").astype(np.int32)


def _ring_worker(rank, world, port, out, kind, epochs, rings=1, n_items=50, redeal_every=1, n_train=None):
    from cornac_amd.dist import BinConveyorBprTrainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k = 6
        n_train = n_items if n_train is None else n_train
        indptr, indices = _ring_data(rank, n_items=n_train)
        log = []
        if kind == "mark":
            factory = lambda *a: _ConveyorMarkTrainer(rank, log, *a)
        else:
            factory = lambda *a: _ConveyorOracleTrainer(*a)
        ring = BinConveyorBprTrainer(indptr, indices, len(indptr) - 1, n_items, k, torch.device("cpu"), trainer_factory=factory, seed=5,
                                     rings=rings, item_order=_ring_order(world, n_train), redeal_every=redeal_every,
                                     n_train_items=n_train)
        U0, V0, B0 = _ring_tables(len(indptr) - 1, n_items, k, rank)
        ring.load_items(V0, B0)
        ring.set_user_factors(U0)
        for _ in range(epochs):
            ring.run_epoch(0.05, 0.01)
        c, s = ring.finish()
        V, B = ring.gather()
        out[rank] = (V, B, ring.get_user_factors(), list(ring.steps_trained), log, c, s, ring.nnz, ring.redeals,
                     (ring.n_bins, ring.bpb, ring.cap))
        ring.close()
    finally:
        dist.destroy_process_group()


def test_ring_conveyor_world2_every_rank_trains_every_block_once_per_epoch():
    """the bookkeeping of BinConveyorBprTrainer over two gloo ranks: 4 blocks, an epoch = 4 steps, rank r trains block
    (2 r + t) % 4 in step t — every (rank, block) pair exactly once per epoch, a block never in two places; between the epochs
    the rows are re-dealt to new slots (one all_to_all) without losing or duplicating one; after the epochs every rank gathers
    the same table: every train row moved by (1 + 2) per epoch, the rows beyond the train items untouched"""
    out = mp.Manager().dict()
    mp.spawn(_ring_worker, args=(2, _free_port(), out, "mark", 3, 1, 53, 1, 50), nprocs=2, join=True)
    _, V0, B0 = _ring_tables(120, 53, 6, 0)
    for rank in (0, 1):
        V, B, U, steps, log, c, s, nnz, redeals, _ = out[rank]
        assert steps == [(t % 4, (2 * rank + t) % 4) for t in range(12)]
        assert log == [(2 * rank + t) % 4 for t in range(12)] and c == 3 * 50 and redeals == 2
        assert np.allclose(V[:50], V0[:50] + 3 * 3.0) and np.allclose(B[:50], B0[:50] + 3 * 30.0)
        assert np.array_equal(V[50:], V0[50:]) and np.array_equal(B[50:], B0[50:])
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def _serial_conveyor(world, rings, n_items, epochs, k, redeal_every, dims):
    """ONE process running the same (step, rank, ring) triples one after the other on ONE item table in item order: a block's
    buffer is cut out of the table by the layout of the step's epoch, trained, and written back — so the re-deal is implicit"""
    from cornac_amd.dist import ring_strides
    from oracle import oracle as orc

    strides = ring_strides(world, rings)
    K, nb = len(strides), 2 * world
    n_bins, bpb, cap = dims
    W = bpb * cap
    _, V, B = _ring_tables(120, n_items, k, 0)
    V, B = V.copy(), B.copy()
    order = _ring_order(world, n_items)
    Us, trainers = [], []
    for rank in range(world):
        indptr, indices = _ring_data(rank, n_items=n_items)
        U = torch.as_tensor(_ring_tables(len(indptr) - 1, n_items, k, rank)[0].copy())
        tr = _ConveyorOracleTrainer(indptr, indices, len(indptr) - 1, n_items, k, U)
        tr.seed_hogwild((5 * 0x9E3779B97F4A7C15 + 7919 * rank + 1) & 0xFFFFFFFFFFFFFFFF)
        assert tr.conveyor_setup(nb * K, order, (5 ^ 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF) == dims
        Us.append(U)
        trainers.append(tr)
    for t in range(epochs * nb):
        epoch, ts = divmod(t, nb)
        lay = epoch - epoch % redeal_every
        for rank in range(world):
            tr = trainers[rank]
            slot_item, item_slot = tr._layout(lay)
            for g, stride in enumerate(strides):
                p = (rank * pow(stride, -1, world)) % world
                blk = ((2 * p + ts) % nb) * K + g
                items = slot_item[blk * W: (blk + 1) * W]
                ok = items >= 0
                Vb, Bb = np.zeros((W, k), np.float32), np.zeros(W, np.float32)
                Vb[ok], Bb[ok] = V[items[ok]], B[items[ok]]
                tr.train_block(blk, items, item_slot, Vb, Bb, 0.05, 0.01, True)
                V[items[ok]], B[items[ok]] = Vb[ok], Bb[ok]
    return V, B, Us


@pytest.mark.parametrize("world,rings,n_items,redeal_every", [(2, 1, 50, 1), (3, 2, 50, 1), (4, 2, 50, 2), (8, 4, 200, 1)])
def test_ring_conveyor_equals_its_serial_execution(world, rings, n_items, redeal_every):
    """real arithmetic (the oracle's BPR loop per block): the steps of one conveyor step touch disjoint user rows and
    disjoint item blocks, so the gloo ranks — blocks travelling between them, rows re-dealt between the epochs by the
    all_to_all — must produce bit for bit what ONE process gets by running the same (step, rank, ring) triples one after the
    other on one table; and the model learns.  world 3 with two rings: the blocks of ring 1 travel r -> r - 2 (the other
    direction of the links); world 4: strides 1 and 3, a re-deal every second epoch; world 8 with four rings (strides 1, 7,
    3, 5: the configuration meant for configs[4] on a node: 64 blocks, 16 steps per epoch, four blocks trained in one launch
    and four in flight per rank and step)."""
    epochs, k = 4, 6
    out = mp.Manager().dict()
    mp.spawn(_ring_worker, args=(world, _free_port(), out, "oracle", epochs, rings, n_items, redeal_every), nprocs=world, join=True)
    dims = out[0][9]
    assert dims[0] == dims[1] * 2 * world * rings
    V, B, Us = _serial_conveyor(world, rings, n_items, epochs, k, redeal_every, dims)
    for rank in range(world):
        Vr, Br, Ur, steps, _, c, s, n, redeals, _ = out[rank]
        assert 0 < c and c + s <= epochs * n and redeals == (epochs - 1) // redeal_every
        assert len(steps) == epochs * 2 * world * rings and len(set(steps[: 2 * world * rings])) == 2 * world * rings, "every block once per epoch"
        assert np.array_equal(Vr, V) and np.array_equal(Br, B), "rank %d: the conveyor's table differs from the serial execution" % rank
        assert np.array_equal(Ur, Us[rank].numpy())
        indptr, indices = _ring_data(rank, n_items=n_items)
        assert _pairwise_accuracy(Ur, Vr, Br, indptr, indices, n_items) > (0.62 if n_items <= 50 else 0.55)


def test_conveyor_blocks_are_redealt_so_every_item_pair_can_meet():
    """round 5's conveyor fixed item i to block i % 2NK for the whole fit: 1 - 1 / 2NK of all (positive, negative) item pairs
    could never be drawn (recom_bpr.pyx:235-238 draws j over ALL items).  Here a block is a range of bins of the epoch's deal:
    over the epochs' layouts every pair of items shares a BIN at the rate 1 / n_bins (and a block at 1 / n_blocks), whatever
    their ids or popularity ranks; with the layout frozen (the static conveyor's behaviour) most pairs never meet."""
    from oracle import oracle as orc

    n_items, n_blocks, n_bins = 26744, 64, 448      # ML-20M's items in 64 blocks (N = 8, four rings) of 7 bins
    order = np.random.RandomState(1).permutation(n_items).astype(np.int32)
    cap = -(-n_items // n_bins)
    rs = np.random.RandomState(2)
    x, y = rs.randint(0, n_items, 4000), rs.randint(0, n_items, 4000)
    x, y = x[x != y], y[x != y]
    epochs = 1500
    same_bin, same_block = np.zeros(len(x), np.int64), np.zeros(len(x), np.int64)
    for e in range(epochs):
        _, item_slot = orc.ldsbin_layout(77, e, n_bins, n_items, order)
        bx, by = item_slot[x] // cap, item_slot[y] // cap
        same_bin += bx == by
        same_block += (bx // (n_bins // n_blocks)) == (by // (n_bins // n_blocks))
    assert abs(same_bin.mean() / epochs * n_bins - 1.0) < 0.1 and abs(same_block.mean() / epochs * n_blocks - 1.0) < 0.05
    assert (same_block == 0).mean() < 1e-3 and (same_bin == 0).mean() < 0.08    # P(never in 1500 epochs) = e^-3.3 = 0.035 per pair
    assert 0.7 <= same_bin.var() / same_bin.mean() <= 1.4
    _, frozen = orc.ldsbin_layout(77, 0, n_bins, n_items, order)                   # negative control: one layout for ever
    never = (frozen[x] // cap // (n_bins // n_blocks)) != (frozen[y] // cap // (n_bins // n_blocks))
    assert abs(never.mean() - (1 - 1 / n_blocks)) < 0.01


def test_ring_strides():
    from cornac_amd.dist import ring_strides

    assert ring_strides(8, 4) == [1, 7, 3, 5] and ring_strides(8, 1) == [1] and ring_strides(8, 9) == [1, 7, 3, 5]
    assert ring_strides(2, 4) == [1] and ring_strides(1, 3) == [1, 1, 1] and ring_strides(3, 2) == [1, 2] and ring_strides(6, 4) == [1, 5]


def _fit_ring_worker(rank, world, port, out):
    import cornac_amd as ca
    from cornac_amd.dist import fit_bpr_ring, fit_bpr_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = ca.Dataset.from_uir(_model_data(), seed=3)
        bpr = ca.BPR(k=6, max_iter=8, learning_rate=0.05, lambda_reg=0.01, seed=rank, mode="hogwild")
        seen = []

        def factory(*a):
            seen.append(_ConveyorOracleTrainer(*a))
            return seen[-1]

        fit_bpr_ring(bpr, ds, trainer_factory=factory)
        out[rank] = dict(U=bpr.u_factors.copy(), V=bpr.i_factors.copy(), B=bpr.i_biases.copy(), stats=bpr.fit_stats,
                         order=seen[0].rank_item.copy(), deal_seed=seen[0].deal_seed)
        with pytest.raises(ValueError):
            fit_bpr_ring(ca.BPR(k=4, seed=1), ds)   # seeded => sequential semantics do not shard
        with pytest.raises(ValueError):               # WBPR with the global popularity is regime 1's
            fit_bpr_ring(ca.WBPR(k=4, max_iter=1, mode="hogwild", seed=rank), ds, trainer_factory=factory, local_popularity=False)
    finally:
        dist.destroy_process_group()


def test_model_level_ring_fit_returns_one_complete_model_on_every_rank():
    """fit_bpr_ring on two gloo ranks (blocks rotating and re-dealt, real arithmetic behind the trainers): both ranks return
    the SAME complete model, bit for bit — there is no replica to reconcile —, started from rank 0's tables, dealt with ONE
    key over the popularity order of the whole matrix (the models' own seeds differ), and it ranks the training positives
    above random negatives"""
    import cornac_amd as ca

    out = mp.Manager().dict()
    mp.spawn(_fit_ring_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = out[0], out[1]
    for name in ("U", "V", "B", "order"):
        assert np.array_equal(a[name], b[name]), name
    assert a["stats"] == b["stats"] and a["stats"][0][0] > 0 and a["deal_seed"] == b["deal_seed"]
    ds = ca.Dataset.from_uir(_model_data(), seed=3)
    X = ds.matrix
    assert np.array_equal(a["order"], np.argsort(-np.bincount(X.indices, minlength=ds.num_items), kind="stable"))
    assert _pairwise_accuracy(a["U"], a["V"], a["B"], X.indptr, X.indices, ds.num_items) > 0.7


def _pop_worker(rank, world, port, out):
    from cornac_amd.dist import global_negative_population

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, indices, n_items = _popularity_data(rank)
        out[rank] = (global_negative_population(indices, n_items), global_negative_population(indices, n_items, at_most=500))
    finally:
        dist.destroy_process_group()


def test_global_negative_population_world2():
    """WBPR over ranks (recom_wbpr.pyx:135): every rank builds the population of the WHOLE matrix from the all-reduced item
    degrees — the same multiset on both ranks, item i exactly degree(i) times; scaled down it keeps the proportions and
    every interacted item"""
    out = mp.Manager().dict()
    mp.spawn(_pop_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    deg = sum(np.bincount(_popularity_data(r)[1], minlength=90) for r in (0, 1))
    for r in (0, 1):
        full, small = out[r]
        assert np.array_equal(np.bincount(full, minlength=90), deg)
        hs = np.bincount(small, minlength=90)
        assert len(small) <= 600 and ((hs > 0) == (deg > 0)).all()
        assert np.abs(hs / hs.sum() - deg / deg.sum()).max() < 0.01
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])



# ---- MF block rotation (regime 2 for MF): the ranks' parallel run == the serial execution of the same steps --------------
class _OracleMfBlockTrainer:
    """stands in for dist._DeviceMfBlockTrainer with real MF arithmetic: the oracle's sequential fit_sgd loop
    (backend_cpu.pyx:56-90) over the block's ratings, IN PLACE on the rank's user tables and on the buffer that holds the block"""

    def __init__(self, rid, lid, val, n_users, rows, k, U, Bu):
        from oracle import oracle as orc

        self.orc = orc
        self.rid, self.lid = np.ascontiguousarray(rid, np.int64), np.ascontiguousarray(lid, np.int64)
        self.val = np.ascontiguousarray(val, np.float32)
        self.U, self.Bu, self.k, self.sq = U.numpy(), Bu.numpy(), k, 0.0

    def enqueue(self, V, Bi, lr, reg, mu, use_bias):
        loss = np.zeros(1, np.float32)
        v, bi = V.numpy(), Bi.numpy()
        assert v.flags.c_contiguous and bi.flags.c_contiguous
        self.orc.lib().oracle_mf_fit(self.rid.copy(), self.lid.copy(), self.val.copy(), len(self.val), self.U, v, self.Bu, bi, self.k,
                                     lr, reg, mu, 1, 1, int(use_bias), 0, loss.ctypes.data)
        self.sq += 2.0 * float(loss[0])

    def sync(self):
        sq, self.sq = self.sq, 0.0
        return sq

    def close(self):
        pass


def _mf_rotation_worker(rank, world, port, out, epochs, n_items):
    from cornac_amd.dist import MfBlockRotationTrainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k = 5
        rid, cid, val, n_users, _ = _mf_data(rank, n_users=60, n_items=n_items, per_user=12)
        order = np.random.RandomState(3).permutation(n_items)
        rot = MfBlockRotationTrainer(rid, cid, val, n_users, n_items, k, torch.device("cpu"), trainer_factory=_OracleMfBlockTrainer,
                                     item_order=order)
        init = np.random.RandomState(7)
        rot.load_items(init.normal(0, 0.05, (n_items, k)).astype(np.float32), init.normal(0, 0.05, n_items).astype(np.float32))
        ur = np.random.RandomState(40 + rank)
        rot.set_user_factors(ur.normal(0, 0.05, (n_users, k)).astype(np.float32), np.zeros(n_users, np.float32))
        sq = []
        for _ in range(epochs):
            rot.run_epoch(0.02, 0.02, 3.2)
            sq.append(rot.finish())
        V, Bi = rot.gather()
        out[rank] = (V, Bi, rot.get_user_factors(), list(rot.steps_trained), sq)
        rot.close()
    finally:
        dist.destroy_process_group()


def _serial_mf_rotation(world, epochs, n_items):
    """the same steps, one rank after the other, on ONE copy of the item table (no blocks, no buffers, no transfers)"""
    from oracle import oracle as orc

    k, nb = 5, 2 * world
    order = np.random.RandomState(3).permutation(n_items)
    pos = np.empty(n_items, np.int64)
    pos[order] = np.arange(n_items)
    init = np.random.RandomState(7)
    V, Bi = init.normal(0, 0.05, (n_items, k)).astype(np.float32), init.normal(0, 0.05, n_items).astype(np.float32)
    data, users, sq = [], [], [[0.0] * epochs for _ in range(world)]
    for r in range(world):
        rid, cid, val, n_users, _ = _mf_data(r, n_users=60, n_items=n_items, per_user=12)
        ur = np.random.RandomState(40 + r)
        data.append((rid, cid, val))
        users.append((ur.normal(0, 0.05, (n_users, k)).astype(np.float32), np.zeros(n_users, np.float32)))
    for e in range(epochs):
        for t in range(nb):
            for r in range(world):
                rid, cid, val = data[r]
                sel = np.flatnonzero(pos[cid] % nb == (2 * r + t) % nb)
                if len(sel) == 0:
                    continue
                loss = np.zeros(1, np.float32)
                orc.lib().oracle_mf_fit(rid[sel].copy(), cid[sel].copy(), val[sel].copy(), len(sel), users[r][0], V, users[r][1], Bi, k,
                                        0.02, 0.02, 3.2, 1, 1, 1, 0, loss.ctypes.data)
                sq[r][e] += 2.0 * float(loss[0])
    return V, Bi, users, sq


@pytest.mark.parametrize("world,n_items", [(2, 41), (3, 50)])
def test_mf_block_rotation_equals_its_serial_execution(world, n_items):
    """MfBlockRotationTrainer over gloo ranks with REAL MF arithmetic (the oracle's fit_sgd loop on every (rank, block) step):
    rank r trains block (2 r + t) mod 2 N in step t, blocks travel to rank r - 1 — and the result is BIT FOR BIT what one
    process gets executing the same steps one after the other on one table: every rating applied exactly once per epoch to the
    one copy of its item row (backend_cpu.pyx:62-88), nothing reconciled.  n_items not a multiple of 2 N: ragged blocks."""
    out, epochs = mp.Manager().dict(), 3
    mp.spawn(_mf_rotation_worker, args=(world, _free_port(), out, epochs, n_items), nprocs=world, join=True)
    V, Bi, users, sq = _serial_mf_rotation(world, epochs, n_items)
    for r in range(world):
        Vr, Bir, (Ur, Bur), steps, sq_r = out[r]
        assert np.array_equal(Vr, V) and np.array_equal(Bir, Bi)            # every rank gathers the same, serial, table
        assert np.array_equal(Ur, users[r][0]) and np.array_equal(Bur, users[r][1])
        assert steps == [(t, (2 * r + t) % (2 * world)) for _ in range(epochs) for t in range(2 * world)]
        assert np.allclose(sq_r, sq[r], rtol=1e-6)
    assert sq[0][-1] < sq[0][0]                                             # and it learns


def _fit_rotation_worker(rank, world, port, out):
    import cornac_amd as ca
    from cornac_amd.dist import fit_mf_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = ca.Dataset.from_uir(_model_data(), seed=3)
        mf = ca.MF(k=6, max_iter=15, learning_rate=0.02, lambda_reg=0.02, seed=rank, mode="hogwild")
        fit_mf_sharded(mf, ds, regime="rotation", block_trainer_factory=_OracleMfBlockTrainer)
        out[rank] = dict(mU=mf.u_factors.copy(), mV=mf.i_factors.copy(), mBu=mf.u_biases.copy(), mBi=mf.i_biases.copy(),
                         mloss=mf.loss_history.copy(), mu=float(mf.global_mean))
        with pytest.raises(ValueError):
            fit_mf_sharded(ca.MF(k=6, seed=1, mode="hogwild"), ds, regime="sideways")
    finally:
        dist.destroy_process_group()


def test_model_level_mf_block_rotation_returns_one_complete_model_on_every_rank():
    """fit_mf_sharded(regime="rotation") on two gloo ranks with real MF arithmetic behind the block handles: both ranks return
    the same complete model (user rows of both ranges gathered, the item side gathered from the blocks' homes), its loss falls
    every epoch, and it fits the data better than the mean."""
    import cornac_amd as ca

    out = mp.Manager().dict()
    mp.spawn(_fit_rotation_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = out[0], out[1]
    for name in ("mU", "mV", "mBu", "mBi"):
        assert np.array_equal(a[name], b[name]), name
    assert np.allclose(a["mloss"], b["mloss"]) and a["mloss"][-1] < 0.85 * a["mloss"][0] and np.all(np.diff(a["mloss"]) < 0)
    ds = ca.Dataset.from_uir(_model_data(), seed=3)
    rid, cid, val = ds.uir_tuple
    pred = a["mu"] + a["mBu"][rid] + a["mBi"][cid] + np.einsum("nk,nk->n", a["mU"][rid], a["mV"][cid])
    assert float(np.sqrt(np.mean((pred - val) ** 2))) < 0.9 * float(np.sqrt(np.mean((a["mu"] - val) ** 2)))
    for lo, hi in ((0, 20), (ds.num_users - 20, ds.num_users)):
        assert np.abs(a["mBu"][lo:hi]).max() > 1e-3
