"""Multi-GPU BPR: one process per GPU, users partitioned across ranks.  Two regimes for the item table
(SURVEY.md §8e): (1) replicated and reconciled with RCCL all-reduce of its deltas — `ShardedBprTrainer`,
described first; (2) sharded by row with all-to-all exchanges of the touched rows — `RowShardedBprTrainer`.

  * every rank owns a disjoint user population (its CSR slice and its U rows) -> user rows never
    leave the GPU and never conflict across GPUs: no data-path collective for them;
  * V and B are replicated; every `sync_every` samples each rank all-reduces its local delta
    (V - V_base, B - B_base) together with the rows it touched as ONE flat fp32 bucket and rebases: rows one
    rank touched receive that rank's SGD steps, rows c ranks touched the sum of their steps / sqrt(c)
    (see ItemTableReplica) — bounded-delay asynchrony, the same class as Hogwild;
  * RCCL runs over xGMI via torch.distributed (backend "nccl"); on CPU-only hosts the same code
    path runs over gloo with a host stand-in for the trainer (tests/test_dist_cpu.py).

The trainer kernels, the delta computation and the collective all run on ONE dedicated torch stream
(handed to the library with cornac_hip_bpr_set_stream), so chunk -> delta -> all-reduce -> rebase ->
next chunk is stream-ordered without host syncs.  (torch's default stream is the NULL stream, which
the C ABI reserves for "use the handle's own stream" — hence the explicit side stream.)
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist


class ItemTableReplica:
    """Flat [V | B] buffer + base copy + exchange of the ranks' deltas.

    Reconciliation rule: a row's new value is  base + (sum over ranks of the row's delta) / sqrt(c),  c = number of
    ranks that touched the row since the last exchange.  Rows one rank touched keep that rank's SGD steps
    unchanged.  For rows every rank touched (the popular items) plain summation applies R stale copies of nearly
    the same gradient and diverges with growing R; the plain average is stable but discounts the item side to one
    rank's worth of progress per epoch; 1/sqrt(c) with >= 16 exchanges per epoch keeps the consolidated model
    within 0.01-0.02 pairwise accuracy of a single rank's at R = 2, 4, 8 (tools/emulate_ranks.py)."""

    def __init__(self, total_items, k, device, group=None, trainer=None):
        self.total_items, self.k = int(total_items), int(k)
        n = self.total_items * self.k + self.total_items
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.base = torch.zeros(n, dtype=torch.float32, device=device)
        self.group = group
        self.trainer = trainer  # on a GPU the two elementwise passes are fused HIP kernels of libcornac_hip
        self._pending = None

    @property
    def V(self):
        return self.flat[: self.total_items * self.k].view(self.total_items, self.k)

    @property
    def B(self):
        return self.flat[self.total_items * self.k:]

    def load(self, V, B):
        self.V.copy_(torch.as_tensor(np.ascontiguousarray(V)))
        self.B.copy_(torch.as_tensor(np.ascontiguousarray(B)))
        self.base.copy_(self.flat)

    def sync(self):
        """blocking form of begin_sync + finish_sync"""
        self.begin_sync()
        self.finish_sync()

    # Overlapped form: the all-reduce of chunk c's delta runs (on RCCL's stream) while chunk c+1 trains.
    #   begin_sync:   d = flat - base (local updates since the last rebase) and the rows it touched; keep a copy of d;
    #                 all-reduce [d | touched] asynchronously (one bucket)
    #   finish_sync:  R = sum(d) / sqrt(max(sum(touched), 1)) per row arrived -> flat += R - d_local, base += R
    # After finish_sync, flat - base is exactly the local delta accumulated since begin_sync, so the next
    # begin_sync sends only new work; other ranks' updates reach a replica one chunk later than with sync().
    def begin_sync(self):
        assert self._pending is None, "finish_sync() the previous exchange first"
        if not (dist.is_available() and dist.is_initialized()):
            self._pending = (None, None, None)
            return
        n, k = self.total_items, self.k
        bucket = torch.empty(n * k + 3 * n, dtype=torch.float32, device=self.flat.device)
        delta = bucket[: n * k + n]
        if self.trainer is not None and self.flat.is_cuda:
            local = torch.empty(n * k + n, dtype=torch.float32, device=self.flat.device)
            self.trainer.table_delta_begin(self.flat.data_ptr(), self.base.data_ptr(), n, k, bucket.data_ptr(),
                                           local.data_ptr())
        else:
            torch.sub(self.flat, self.base, out=delta)
            bucket[n * k + n: n * k + 2 * n] = (delta[: n * k].view(n, k) != 0).any(dim=1)   # V rows touched
            bucket[n * k + 2 * n:] = delta[n * k:] != 0                                         # biases touched
            local = delta.clone()
        work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending = (work, bucket, local)

    def step_sync(self):
        """finish_sync() of the pending exchange followed by begin_sync() of the next one; on a GPU the two table
        passes are one fused kernel"""
        if self._pending is None or self._pending[0] is None or self.trainer is None or not self.flat.is_cuda:
            self.finish_sync()
            self.begin_sync()
            return
        work, bucket_prev, local_prev = self._pending
        self._pending = None
        work.wait()
        n, k = self.total_items, self.k
        bucket = torch.empty(n * k + 3 * n, dtype=torch.float32, device=self.flat.device)
        local = torch.empty(n * k + n, dtype=torch.float32, device=self.flat.device)
        self.trainer.table_delta_step(self.flat.data_ptr(), self.base.data_ptr(), bucket_prev.data_ptr(),
                                      local_prev.data_ptr(), n, k, bucket.data_ptr(), local.data_ptr())
        work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending = (work, bucket, local)

    def finish_sync(self):
        if self._pending is None:
            return
        work, bucket, local = self._pending
        self._pending = None
        if work is None:
            self.base.copy_(self.flat)
            return
        work.wait()  # stream-level wait on CUDA, blocking on gloo
        n, k = self.total_items, self.k
        if self.trainer is not None and self.flat.is_cuda:
            self.trainer.table_delta_finish(self.flat.data_ptr(), self.base.data_ptr(), bucket.data_ptr(),
                                            local.data_ptr(), n, k)
            return
        delta = bucket[: n * k + n]
        delta[: n * k].view(n, k).div_(bucket[n * k + n: n * k + 2 * n].clamp_(min=1.0).sqrt_().unsqueeze(1))
        delta[n * k:].div_(bucket[n * k + 2 * n:].clamp_(min=1.0).sqrt_())
        self.flat.add_(delta - local)
        self.base.add_(delta)


class ShardedBprTrainer:
    """Drives one rank's cornac_hip BPR handle plus the replicated item table."""

    def __init__(self, trainer, total_items, k, device, sync_every, group=None):
        self.trainer = trainer
        self.table = ItemTableReplica(total_items, k, device, group, trainer=trainer if device.type == "cuda" else None)
        self.sync_every = int(sync_every)
        self.device = device
        self.stream = None
        if device.type == "cuda":
            self.stream = torch.cuda.Stream(device)
            if trainer is not None:
                torch.cuda.synchronize(device)
                trainer.bind_device(None, self.table.V.data_ptr(), self.table.B.data_ptr())
                trainer.set_stream(self.stream.cuda_stream)

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def load_items(self, V, B):
        with self._on_stream():
            self.table.load(V, B)
        if self.stream is not None:
            self.stream.synchronize()

    def run(self, n_samples, lr, reg, use_bias=True, neg_population=0, flags=0):
        """enqueue n_samples hogwild samples in sync_every-sized chunks; the item-table exchange of chunk c is
        in flight while chunk c+1 trains (ItemTableReplica.begin_sync / finish_sync)"""
        left = int(n_samples)
        with self._on_stream():
            while left > 0:
                n = min(left, self.sync_every)
                self.trainer.hogwild_enqueue(n, lr, reg, use_bias, neg_population, flags)
                self.table.step_sync()   # finish chunk c-1's exchange (its all-reduce overlapped this launch), begin c's
                left -= n

    def finish(self):
        with self._on_stream():
            self.table.finish_sync()
        out = self.trainer.sync()
        if self.stream is not None:
            self.stream.synchronize()
        return out


class RowShardedItemTable:
    """Item factors and biases sharded by row (SURVEY.md §8e regime 2): item i lives on rank i % N at local row
    i // N.  `fetch` pulls a de-duplicated set of rows from their owners, `push` returns deltas to them:

        ids  --all_to_all-->  owners            (int32 local rows, variable split sizes)
        rows <--all_to_all--  owners gather     ([n, k] fp32 + [n] bias)
        ...  local BPR updates on the staged rows ...
        deltas --all_to_all--> owners scatter-add (atomic: the same row may come from several ranks)

    Row gather / scatter-add are HIP kernels of libcornac_hip reached through `ops` (`gather(table, ids, out)`,
    `scatter_add(table, ids, delta)`); RCCL moves the data (torch.distributed, backend nccl).  Requests are
    addressed in "owner-major" order g(i) = (i % N) * rows_per_rank + i // N, so a sorted unique list of g is
    already bucketed by owner."""

    def __init__(self, total_items, k, device, ops, group=None):
        self.total_items, self.k, self.device, self.ops, self.group = int(total_items), int(k), device, ops, group
        on = dist.is_available() and dist.is_initialized()
        self.collective = on  # a size-1 group still goes through RCCL (exercised by the single-GPU tests)
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.rows_per_rank = (self.total_items + self.world - 1) // self.world
        self.V = torch.zeros(self.rows_per_rank, self.k, dtype=torch.float32, device=device)
        self.B = torch.zeros(self.rows_per_rank, dtype=torch.float32, device=device)

    def owner_major(self, item_ids):
        return (item_ids % self.world) * self.rows_per_rank + item_ids // self.world

    def load(self, V, B):
        """keeps this rank's rows of the full host tables"""
        mine = np.arange(self.rank, self.total_items, self.world)
        self.V[: len(mine)].copy_(torch.as_tensor(np.ascontiguousarray(np.asarray(V, np.float32)[mine])))
        self.B[: len(mine)].copy_(torch.as_tensor(np.ascontiguousarray(np.asarray(B, np.float32)[mine])))

    def _exchange(self, send, send_counts, recv_counts):
        if not self.collective:
            return send
        recv = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                               group=self.group)
        return recv

    def dedupe(self, g, valid):
        """De-duplicate the owner-major indices `g` (int64 [m], entries with valid == False ignored) WITHOUT sorting
        and with ONE host synchronisation: a flag per table row, an inclusive scan of the flags (slot of a touched row
        = scan - 1; scan order = owner-major order, so the unique list is bucketed by owner), the per-owner counts read
        off the scan at the owner boundaries, exchanged with the peers on the device and copied to the host together
        with the local counts.  Returns (uniq [n_unique] int64 sorted, slot [m] int32, send_counts, recv_counts)."""
        n_rows = self.world * self.rows_per_rank
        gg = torch.where(valid, g, torch.full_like(g, n_rows))          # ignored entries point at a spare flag
        mark = torch.zeros(n_rows + 1, dtype=torch.int32, device=g.device)
        mark[gg] = 1
        mark[n_rows] = 0
        scan = torch.cumsum(mark, 0, dtype=torch.int32)
        slot = (scan[gg] - 1).clamp_(min=0)
        bounds = torch.arange(1, self.world + 1, device=g.device) * self.rows_per_rank - 1
        ends = scan[bounds].to(torch.int64)
        sc = ends - torch.cat([ends.new_zeros(1), ends[:-1]])
        if self.collective:
            rc = torch.empty_like(sc)
            dist.all_to_all_single(rc, sc, group=self.group)
        else:
            rc = sc
        counts = torch.stack([sc, rc]).tolist()                           # the ONE host synchronisation
        send_counts, recv_counts = counts[0], counts[1]
        uniq = torch.empty(int(sum(send_counts)), dtype=torch.int64, device=g.device)
        if len(uniq):
            keep = torch.where(valid, slot.long(), torch.full_like(g, len(uniq)))   # ignored entries -> a spare slot
            buf = torch.empty(len(uniq) + 1, dtype=torch.int64, device=g.device)
            buf[keep] = gg
            uniq = buf[:-1]
        return uniq, slot, send_counts, recv_counts

    def fetch(self, uniq_g, send_counts=None, recv_counts=None):
        """uniq_g: sorted unique owner-major indices (int64, on `device`).  Returns (rows [n,k], bias [n], plan).
        Without the split sizes (as `dedupe` returns them) they are derived here with two more host syncs."""
        if send_counts is None:
            bounds = torch.arange(self.world + 1, device=uniq_g.device, dtype=uniq_g.dtype) * self.rows_per_rank
            cuts = torch.searchsorted(uniq_g, bounds).tolist()  # host sync: the split sizes of the exchange
            send_counts = [cuts[r + 1] - cuts[r] for r in range(self.world)]
            if self.collective:
                sc = torch.tensor(send_counts, dtype=torch.int64, device=self.device)
                rc = torch.empty_like(sc)
                dist.all_to_all_single(rc, sc, group=self.group)
                recv_counts = rc.tolist()
            else:
                recv_counts = list(send_counts)
        local_rows = (uniq_g % self.rows_per_rank).to(torch.int32)
        wanted = self._exchange(local_rows, send_counts, recv_counts)          # rows the others want from me
        out_rows = torch.empty(len(wanted), self.k, dtype=torch.float32, device=self.device)
        out_bias = torch.empty(len(wanted), 1, dtype=torch.float32, device=self.device)
        self.ops.gather(self.V, wanted, out_rows)
        self.ops.gather(self.B.view(-1, 1), wanted, out_bias)
        rows = self._exchange(out_rows, recv_counts, send_counts)
        bias = self._exchange(out_bias, recv_counts, send_counts).view(-1)
        return rows, bias, (send_counts, recv_counts, wanted)

    def push(self, plan, d_rows, d_bias):
        """owners apply  sum of the received deltas of a row / sqrt(number of ranks that sent one)  (the same
        reconciliation rule as ItemTableReplica: several ranks' stale steps on one popular row are damped)"""
        send_counts, recv_counts, wanted = plan
        got_rows = self._exchange(d_rows.contiguous(), send_counts, recv_counts)
        got_bias = self._exchange(d_bias.contiguous().view(-1, 1), send_counts, recv_counts)
        if self.world > 1 and len(wanted):
            idx = wanted.long()
            senders = torch.zeros(self.rows_per_rank, dtype=torch.float32, device=self.device)
            senders.index_add_(0, idx, torch.ones(len(idx), dtype=torch.float32, device=self.device))
            scale = senders[idx].rsqrt().unsqueeze(1)
            got_rows = got_rows * scale
            got_bias = got_bias * scale
        self.ops.scatter_add(self.V, wanted, got_rows)
        self.ops.scatter_add(self.B.view(-1, 1), wanted, got_bias)

    def gather_full(self):
        """(V [total_items, k], B [total_items]) assembled on every rank (for get_factors / evaluation)"""
        if not self.collective:
            return self.V[: self.total_items].clone(), self.B[: self.total_items].clone()
        Vs = [torch.empty_like(self.V) for _ in range(self.world)]
        Bs = [torch.empty_like(self.B) for _ in range(self.world)]
        dist.all_gather(Vs, self.V, group=self.group)
        dist.all_gather(Bs, self.B, group=self.group)
        V = torch.stack(Vs, 1).reshape(-1, self.k)[: self.total_items]   # row r of rank q -> item r * N + q
        B = torch.stack(Bs, 1).reshape(-1)[: self.total_items]
        return V, B


class DeviceRowOps:
    """row gather / scatter-add through libcornac_hip on the trainer's stream (device tensors only)"""

    def __init__(self, trainer):
        self.trainer = trainer

    def gather(self, table, ids, out):
        self.trainer.gather_rows(table.data_ptr(), ids.data_ptr(), len(ids), table.shape[1], out.data_ptr())

    def scatter_add(self, table, ids, delta):
        self.trainer.scatter_add_rows(table.data_ptr(), ids.data_ptr(), len(ids), table.shape[1], delta.data_ptr())


class RowShardedBprTrainer:
    """One rank of BPR with the item table sharded by row.  Users (CSR slice, U rows) are rank-local as in
    regime 1; each micro-batch samples triplets, fetches the touched item rows, applies the hogwild update on
    the staged rows and pushes the deltas back.  Between a fetch and the matching push other ranks may fetch the
    same rows: the usual bounded-staleness asynchrony, one micro-batch deep."""

    BIAS_STRIDE = 32  # staged biases sit one per 128-byte line (see DESIGN.md "padded bias table")

    def __init__(self, trainer, total_items, k, device, micro_batch, group=None, ops=None):
        self.trainer, self.device, self.micro_batch, self.group = trainer, device, int(micro_batch), group
        self.stream = None
        if device.type == "cuda":
            self.stream = torch.cuda.Stream(device)
            torch.cuda.synchronize(device)
            trainer.set_stream(self.stream.cuda_stream)
        with self._on_stream():
            self.table = RowShardedItemTable(total_items, k, device, ops or DeviceRowOps(trainer), group)
        if device.type == "cuda":
            # the handle's own (full-size) item table is not used in this mode: release it
            trainer.bind_device(None, self.table.V.data_ptr(), self.table.B.data_ptr())
        self.rows_fetched = 0
        self._valid_draws = torch.zeros((), dtype=torch.int64, device=device)

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    @property
    def triplets(self):
        """valid (not skipped) draws applied so far"""
        return int(self._valid_draws.item())

    def load_items(self, V, B):
        with self._on_stream():
            self.table.load(V, B)
        if self.stream is not None:
            self.stream.synchronize()

    def _sample(self, n):
        u = torch.empty(n, dtype=torch.int32, device=self.device)
        i, j = torch.empty_like(u), torch.empty_like(u)
        self.trainer.sample_triplets(n, u.data_ptr(), i.data_ptr(), j.data_ptr())
        return u, i, j

    def _apply(self, u, slot_i, slot_j, rows, bias_pad, lr, reg, use_bias):
        self.trainer.apply_triplets(u.data_ptr(), slot_i.data_ptr(), slot_j.data_ptr(), len(u), rows.data_ptr(),
                                    bias_pad.data_ptr(), self.BIAS_STRIDE, lr, reg, use_bias)

    def run(self, n_samples, lr, reg, use_bias=True):
        """n_samples draws on this rank, in micro-batches.  Every micro-batch is a collective (three all-to-alls), so
        all ranks must go through the same number of them: the ranks agree on the maximum up front and a rank whose
        user shard runs out of draws keeps serving the others' fetches / pushes with empty batches."""
        left = int(n_samples)
        rounds = (left + self.micro_batch - 1) // self.micro_batch
        if self.table.collective and self.table.world > 1:
            r = torch.tensor([rounds], dtype=torch.int64, device=self.device)
            with self._on_stream():
                dist.all_reduce(r, op=dist.ReduceOp.MAX, group=self.group)
            rounds = int(r.item())
        with self._on_stream():
            for _ in range(rounds):
                n = min(left, self.micro_batch)
                left -= n
                if n > 0:
                    u, i, j = self._sample(n)
                else:   # nothing left to draw here: an empty batch (all skipped) keeps the collectives matched
                    u = torch.full((0,), -1, dtype=torch.int32, device=self.device)
                    i, j = u.clone(), u.clone()
                valid = u >= 0                                      # skipped draws carry -1 and stay in place: the
                nv = len(u)                                         # apply kernel ignores them (no compaction, no sync)
                g = torch.cat([self.table.owner_major(i.long()), self.table.owner_major(j.long())])
                uniq, slot, send_counts, recv_counts = self.table.dedupe(g, torch.cat([valid, valid]))
                rows, bias, plan = self.table.fetch(uniq, send_counts, recv_counts)
                rows0, bias0 = rows.clone(), bias
                bias_pad = torch.zeros(len(bias), self.BIAS_STRIDE, dtype=torch.float32, device=self.device)
                bias_pad[:, 0] = bias
                if nv and len(uniq):
                    self._apply(u, slot[:nv].contiguous(), slot[nv:].contiguous(), rows, bias_pad, lr, reg, use_bias)
                self.table.push(plan, rows - rows0, bias_pad[:, 0] - bias0)
                self.rows_fetched += len(uniq)
                self._valid_draws = self._valid_draws + valid.sum()   # stays on the device: read through .triplets

    def finish(self):
        out = self.trainer.sync()
        if self.stream is not None:
            self.stream.synchronize()
        return out


def partition_users_by_nnz(indptr, world_size):
    """contiguous user ranges with (almost) equal interaction counts: returns world_size+1 boundaries"""
    indptr = np.asarray(indptr, dtype=np.int64)
    nnz = indptr[-1]
    targets = (np.arange(1, world_size) * nnz) // world_size
    cuts = np.searchsorted(indptr, targets, side="left")
    return np.concatenate([[0], cuts, [len(indptr) - 1]]).astype(np.int64)


def slice_csr(indptr, indices, u0, u1):
    indptr = np.asarray(indptr)
    a, b = int(indptr[u0]), int(indptr[u1])
    return (indptr[u0:u1 + 1] - a).astype(np.int32), np.asarray(indices[a:b], dtype=np.int32)


def rank_users_sharded(rank_fn, users, k, device=None, group=None):
    """Multi-GPU scoring: ranking shards by user block with no data-path collective — every
    rank holds the (replicated or gathered) factor tables, ranks its contiguous block of `users` with
    `rank_fn(block) -> (items [n, k] int32, scores [n, k] float32)` (e.g. `model.rank_batch`, the fused MFMA top-k
    kernel) — followed by one all-gather of the top-k lists.  Returns the full `(items, scores)` for `users` on
    every rank.  Without an initialised process group it is a plain call."""
    users = np.asarray(users)
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    cuts = np.linspace(0, len(users), world + 1).astype(np.int64)
    mine = users[cuts[rank]:cuts[rank + 1]]
    if len(mine):
        items, scores = rank_fn(mine)
        items, scores = np.asarray(items, np.int32), np.asarray(scores, np.float32)
    else:
        items, scores = np.empty((0, k), np.int32), np.empty((0, k), np.float32)
    if world == 1:
        return items, scores
    width = items.shape[1] if len(mine) else k
    cap = int((cuts[1:] - cuts[:-1]).max())
    dev = device if device is not None else torch.device("cpu")
    pad_i = torch.full((cap, width), -1, dtype=torch.int32, device=dev)
    pad_s = torch.full((cap, width), float("-inf"), dtype=torch.float32, device=dev)
    pad_i[: len(mine)] = torch.as_tensor(items)
    pad_s[: len(mine)] = torch.as_tensor(scores)
    all_i = [torch.empty_like(pad_i) for _ in range(world)]
    all_s = [torch.empty_like(pad_s) for _ in range(world)]
    dist.all_gather(all_i, pad_i, group=group)
    dist.all_gather(all_s, pad_s, group=group)
    out_i = np.concatenate([all_i[r][: cuts[r + 1] - cuts[r]].cpu().numpy() for r in range(world)])
    out_s = np.concatenate([all_s[r][: cuts[r + 1] - cuts[r]].cpu().numpy() for r in range(world)])
    return out_i, out_s
