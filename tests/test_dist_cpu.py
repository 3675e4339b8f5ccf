"""world_size-2 gloo tests (CPU) of the multi-GPU path's host logic (cornac_amd/dist.py): user
partitioning and the item-table delta exchange (sum of the deltas / sqrt of the number of touching ranks, per row).  The HIP trainer is replaced by a host stand-in
that applies known per-rank updates, so the reduction algebra is checked exactly."""
import os
import socket

import numpy as np
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


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        sh = ShardedBprTrainer(None, total_items=6, k=4, device=dev, sync_every=100)
        sh.trainer = _FakeTrainer(sh.table, rank)
        V0 = np.arange(24, dtype=np.float32).reshape(6, 4)
        sh.load_items(V0, np.zeros(6, np.float32))
        sh.run(250, lr=0.01, reg=0.0)  # chunks of 100, 100, 50 -> 3 syncs
        correct, _ = sh.finish()
        out[rank] = (sh.table.V.numpy().copy(), sh.table.B.numpy().copy(), sh.trainer.calls, correct)
    finally:
        dist.destroy_process_group()


def test_item_table_allreduce_of_deltas_world2():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    (V_a, B_a, calls_a, c_a), (V_b, B_b, calls_b, c_b) = out[0], out[1]
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
