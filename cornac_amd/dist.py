"""Multi-GPU BPR (and MF, `ShardedMfTrainer`): one process per GPU, users partitioned across ranks.  Two regimes for the item table
(SURVEY.md §8e): (1) replicated and reconciled with RCCL all-reduce of its deltas — `ShardedBprTrainer`,
described first; (2) sharded by row — `BinConveyorBprTrainer`: the table's blocks (bin ranges of the epoch's LDS-bin deal) rotate
over the ranks on point-to-point xGMI hops and are re-dealt at the epoch boundaries; round 2's all-to-all exchange of the touched
rows is kept as `RowShardedBprTrainer`.

  * every rank owns a disjoint user population (its CSR slice and its U rows) -> user rows never
    leave the GPU and never conflict across GPUs: no data-path collective for them;
  * V and B are replicated; every `sync_every` samples each rank all-reduces its local delta
    (V - V_base, B - B_base) together with the rows it touched as ONE flat fp32 bucket and rebases: rows one
    rank touched receive that rank's SGD steps, rows c ranks touched the sum of their steps / sqrt(c)
    (see ItemTableReplica) — bounded-delay asynchrony, the same class as Hogwild;
  * RCCL runs over xGMI via torch.distributed (backend "nccl"); on CPU-only hosts the same code
    path runs over gloo with a host stand-in for the trainer (tests/test_dist_cpu.py).

Two protocols drive regime 1 (ShardedBprTrainer.run_epoch picks): CHUNK LAUNCHES with the overlapped exchange between
them (begin_sync / finish_sync / step_sync), and — where the handle takes the LDS-bin form — the RESIDENT EXCHANGE: one
launch per epoch whose workgroups publish their rows' deltas at the exchange points and apply the landed sums
themselves, fed from a communication stream (run_epoch_resident; csrc/bpr_ldsbin.inc).  exchange_schedule() says how
often and by which rule the replicas are reconciled.

The trainer kernels, the delta computation and the collective all run on ONE dedicated torch stream
(handed to the library with cornac_hip_bpr_set_stream), so chunk -> delta -> all-reduce -> rebase ->
next chunk is stream-ordered without host syncs.  (torch's default stream is the NULL stream, which
the C ABI reserves for "use the handle's own stream" — hence the explicit side stream.)
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist


RULES = {"sqrt": 0, "align": 1}


def exchanges_per_epoch(nnz_per_rank, total_items, updates_per_row=93.5, at_most=64):
    """Item-table exchanges per epoch of the replicated regime.  What makes a stale replica harmful is the number of
    updates a row collects on a rank between two exchanges, not the fraction of an epoch that passes: the emulations
    behind the default (tools/emulate_ranks.py) found 16 exchanges per epoch necessary and sufficient at the
    ML-20M shape with up to 8 ranks — 2 x 20 000 263 / 26 744 / 16 = 93.5 item-row updates per row and exchange.  The rule
    keeps that staleness: the ML-20M shape gets its 16, the configs[4] slice (62.5 M interactions per rank over 10 M item
    rows: 12.5 updates per row and EPOCH) one exchange per epoch, overlapped with the next epoch."""
    x = 2.0 * float(nnz_per_rank) / (float(total_items) * float(updates_per_row))
    return int(min(max(1, int(x + 0.5)), at_most))


def prefers_conveyor(nnz_per_rank, total_items, updates_per_row=93.5):
    """True for SPARSE item sides — a row collects fewer than ~93.5 updates on a rank in a whole epoch (the configs[4] slice: 12.5)
    — where replicas reconciled once per epoch lag in mid-training (0.869 against 0.990 for one process, device measurement,
    profiles/r05_virtual_ranks.log) and the conveyor does not (0.990): fit_bpr_sharded(regime="auto") takes regime 2 there."""
    return 2.0 * float(nnz_per_rank) / (float(total_items) * float(updates_per_row)) < 0.75


def exchange_schedule(nnz_per_rank, total_items, updates_per_row=93.5, at_most=64, max_epochs=1):
    """(exchanges per epoch, epochs per exchange, rule) of the replicated regime for BPR.

    Dense item sides (the ML-20M shape: 1 496 item-row updates per row and epoch on a rank) exchange several times per
    epoch — exchanges_per_epoch() — under the "sqrt" rule (device emulation at R = 8: sqrt 0.673 vs align 0.667 of 0.679
    at 16 exchanges per epoch; below that align is the better one: 0.650 vs 0.604 at 8).  SPARSE item sides (the
    configs[4] slice: 12.5 updates per row and epoch) were emulated on the CPU with the oracle's arithmetic
    (tools/emulate_exchange_interval.py, 8 ranks, profiles/r04_emulate_exchange_interval.log): there every rank touches
    every popular row between two exchanges with a delta that is a sizeable part of the way to its optimum, and "sqrt"
    (sum / sqrt(8) = 2.8 x one rank's step) overshoots — consolidated accuracy 0.753 / 0.668 / 0.817 / 0.487 at one exchange
    every 1 / 2 / 4 / 8 epochs, erratic — while "align" (the mean of aligned deltas, the sum of orthogonal ones) gives
    0.825 / 0.819 / 0.814 / 0.813 against 0.836 for one process on all ranks' data.  So: fewer than 16 exchanges per
    epoch -> "align"; and a rank whose rows collect less than the 93.5 updates per exchange in a whole epoch exchanges
    every floor(93.5 / updates per row and epoch) epochs, at most every `max_epochs` (0.011 below exchanging every
    epoch in the emulation, for a quarter of the table passes).

    Round 5 re-measured the sparse case ON THE DEVICE (tests/test_sharded_gpu.py::test_eight_virtual_ranks_..., 8 virtual
    ranks at the slice's density, profiles/r05_virtual_ranks.log): the replicas reach the single process's quality only at
    convergence (0.9948 vs 0.9964 after 32 epochs); in the MIDDLE of training the averaged, stale item side lags badly
    (16 epochs: 0.869 / 0.708 / 0.618 at one exchange every 1 / 2 / 4 epochs against 0.990 for one process) — every rank
    moves every row the same way and "align" keeps the mean.  The ring conveyor (BinConveyorBprTrainer, fit_bpr_ring) has
    no such trade: 0.990 in the same measurement.  Sparse item sides should take the ring (prefers_conveyor); since round 6
    the multi-epoch intervals are off by default (max_epochs = 1: a sparse side asked to run regime 1 exchanges once per
    epoch, the least bad of the measured schedules) and remain reachable through max_epochs for experiments."""
    x = 2.0 * float(nnz_per_rank) / (float(total_items) * float(updates_per_row))
    if x >= 0.75:
        per_epoch, epochs = int(min(max(1, int(x + 0.5)), at_most)), 1
    else:
        per_epoch, epochs = 1, int(max(1, min(int(max_epochs), int(1.0 / x))))
    return per_epoch, epochs, ("sqrt" if per_epoch >= 16 else "align")


class ItemTableReplica:
    """Flat [V | B] buffer + base copy + exchange of the ranks' deltas.

    Reconciliation rule "sqrt" (BPR): a row's new value is  base + (sum over ranks of the row's delta) / sqrt(c),
    c = number of ranks that touched the row since the last exchange.  Rows one rank touched keep that rank's SGD steps
    unchanged.  For rows every rank touched (the popular items) plain summation applies R stale copies of nearly
    the same gradient and diverges with growing R; the plain average is stable but discounts the item side to one
    rank's worth of progress per epoch; 1/sqrt(c) with >= 16 exchanges per epoch keeps the consolidated model
    within 0.01-0.02 pairwise accuracy of a single rank's at R = 2, 4, 8 (tools/emulate_ranks.py).

    Rule "align" (MF):  base + S * min(1, sum_r |d_r|^2 / |S|^2),  S = the summed delta of the row, d_r rank r's.  The
    factor is 1 (plain sum) when the ranks' deltas are orthogonal — independent evidence — and 1 / c (their mean) when
    they are c copies of one step: the hot-row case in which sum / sqrt(c) overshoots by sqrt(c) and MF's unbounded
    squared-error steps diverge (round 3: NaN at R = 8 below 16 exchanges per epoch).  Always contractive where the mean
    is, never slower than the mean, a row one rank touched keeps that rank's steps; no exchange count makes it diverge
    in the emulation (tools/emulate_ranks_mf.py --rule align: held-out RMSE 0.445 / 0.434 / 0.431 at 8 / 16 / 32
    exchanges per epoch and R = 8 against 0.428 for one process on all ratings; sqrt: NaN / 0.730 / 0.443).  The bucket's
    per-row slots carry |d_r|^2 instead of the touched flag."""

    def __init__(self, total_items, k, device, group=None, trainer=None, sparse_threshold=None, rule="sqrt"):
        self.total_items, self.k = int(total_items), int(k)
        n = self.total_items * self.k + self.total_items
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.base = torch.zeros(n, dtype=torch.float32, device=device)
        self.group = group
        self.trainer = trainer  # on a GPU the two elementwise passes are fused HIP kernels of libcornac_hip
        if rule not in RULES:
            raise ValueError("rule must be one of %r" % (sorted(RULES),))
        self.rule = rule
        self._pending = None
        # sparse exchange: when at most this fraction of the item rows was touched since the last
        # exchange ON EVERY RANK, the ranks all_gather (row id, delta row) records instead of all-reducing the dense
        # table; None = always dense.  The reconciliation rule is the same, so both forms give the same table.
        self.sparse_threshold = sparse_threshold
        self.exchanges = {"dense": 0, "sparse": 0, "sparse_rows": 0}
        self._count_host = None
        # the dense exchange's buffers, two sets used in turn (exchange c is in flight while c + 1 is being filled): at the
        # configs[4] size a bucket is 5.3 GB — allocated once, not per exchange
        self._sets = [None, None]
        self._turn = 0
        # measurement aid: with a process group of ONE rank RCCL launches nothing; emulate_world = N > 1 puts a stand-in
        # where the all-reduce sits that streams the bytes a ring all-reduce over N ranks moves through a rank
        # (2 (N - 1) / N x the bucket) with a collective's workgroup count (cornac_hip_stream_ring_standin)
        self.emulate_world, self.standin_workgroups = 0, 16
        self._standin = None   # (stream, scratch)
        self._standin_event = None

    def _next_buffers(self):
        """(bucket [n k + 3 n], local [n k + n]) of the next exchange"""
        n, k = self.total_items, self.k
        self._turn ^= 1
        if self._sets[self._turn] is None:
            self._sets[self._turn] = (torch.empty(n * k + 3 * n, dtype=torch.float32, device=self.flat.device),
                                      torch.empty(n * k + n, dtype=torch.float32, device=self.flat.device))
        return self._sets[self._turn]

    def _emulate_collective(self, bucket, stream=None):
        """the stand-in of one all-reduce of `bucket` (see emulate_world); on `stream` (a raw handle: the caller orders it) or
        on a side stream ordered after the current one, whose completion finish_sync() / step_sync() wait for"""
        if self.emulate_world <= 1 or not bucket.is_cuda:
            return
        from . import _lib

        bucket = bucket[(-(bucket.data_ptr() // 4)) % 4:]     # (16-byte accesses: start at an aligned float)
        n = bucket.numel() & ~3
        if n < 4:
            return
        if self._standin is None:
            self._standin = (torch.cuda.Stream(bucket.device), torch.empty(max(4, min(n, 1 << 26)), dtype=torch.float32, device=bucket.device))
        side, scratch = self._standin
        moved = int(2 * (self.emulate_world - 1) * n // self.emulate_world)
        dev = bucket.device.index or 0
        if stream is not None:
            _lib.stream_ring_standin(dev, stream, bucket.data_ptr(), n, scratch.data_ptr(), scratch.numel(), moved, self.standin_workgroups)
            return
        side.wait_stream(torch.cuda.current_stream(bucket.device))
        _lib.stream_ring_standin(dev, side.cuda_stream, bucket.data_ptr(), n, scratch.data_ptr(), scratch.numel(), moved,
                                 self.standin_workgroups)
        self._standin_event = torch.cuda.Event()
        self._standin_event.record(side)

    def _await_emulated(self):
        if self._standin_event is not None:
            torch.cuda.current_stream(self.flat.device).wait_event(self._standin_event)
            self._standin_event = None

    def _factors(self, S, w):
        """per-row factor the summed delta S [m, width] is multiplied with; w = the all-reduced per-row weights"""
        if self.rule == "sqrt":
            return w.clamp(min=1.0).sqrt().reciprocal()
        n2 = (S * S).sum(dim=1) if S.dim() == 2 else S * S
        return torch.where(n2 > 0, (w / n2.clamp(min=1e-38)).clamp(max=1.0), torch.ones_like(n2))

    def _weights(self, dV, dB):
        """this rank's per-row weights of its deltas dV [m, k], dB [m]"""
        if self.rule == "sqrt":
            return (dV != 0).any(dim=1).to(torch.float32), (dB != 0).to(torch.float32)
        return (dV * dV).sum(dim=1), dB * dB

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
        bucket, local = self._next_buffers()
        delta = bucket[: n * k + n]
        if self.trainer is not None and self.flat.is_cuda:
            self.trainer.table_delta_begin(self.flat.data_ptr(), self.base.data_ptr(), n, k, bucket.data_ptr(),
                                           local.data_ptr(), rule=RULES[self.rule])
        else:
            torch.sub(self.flat, self.base, out=delta)
            wV, wB = self._weights(delta[: n * k].view(n, k), delta[n * k:])
            bucket[n * k + n: n * k + 2 * n] = wV   # V rows: touched flag ("sqrt") or |d|^2 ("align")
            bucket[n * k + 2 * n:] = wB             # biases likewise
            local.copy_(delta)
        if self.sparse_threshold is not None and self._begin_sparse(bucket):
            return
        self.exchanges["dense"] += 1
        work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._emulate_collective(bucket)
        self._pending = (work, bucket, local)

    # ---- sparse form: records (row id, [delta V row | delta bias]) of the touched rows, all_gather'ed ----------------
    def _begin_sparse(self, bucket):
        """`bucket` = the dense form's [dV | dB | V-row flags | bias flags].  Returns False (nothing started) when some
        rank touched more than sparse_threshold of the rows: every rank then all-reduces the bucket it already has.
        One device->host copy of ONE count (the max over ranks) sizes the buffers."""
        n, k = self.total_items, self.k
        touched = (bucket[n * k + n: n * k + 2 * n] + bucket[n * k + 2 * n:]) > 0
        cnt = touched.sum().to(torch.int64).reshape(1)
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX, group=self.group)
        if cnt.is_cuda:
            if self._count_host is None:
                self._count_host = torch.empty(1, dtype=torch.int64).pin_memory()
            self._count_host.copy_(cnt, non_blocking=True)
            torch.cuda.current_stream(cnt.device).synchronize()
            cap = int(self._count_host[0])
        else:
            cap = int(cnt[0])
        if cap > self.sparse_threshold * n:
            return False
        world = dist.get_world_size(self.group)
        ids = torch.nonzero(touched).reshape(-1)
        m = int(ids.numel())
        ids_pad = torch.full((max(cap, 1),), -1, dtype=torch.int64, device=self.flat.device)
        rec = torch.zeros((max(cap, 1), k + 1), dtype=torch.float32, device=self.flat.device)
        ids_pad[:m] = ids
        rec[:m, :k] = bucket[: n * k].view(n, k)[ids]
        rec[:m, k] = bucket[n * k: n * k + n][ids]
        all_ids = torch.empty(world * max(cap, 1), dtype=torch.int64, device=self.flat.device)
        all_rec = torch.empty((world * max(cap, 1), k + 1), dtype=torch.float32, device=self.flat.device)
        w1 = dist.all_gather_into_tensor(all_ids, ids_pad, group=self.group, async_op=True)
        w2 = dist.all_gather_into_tensor(all_rec, rec, group=self.group, async_op=True)
        self.exchanges["sparse"] += 1
        self.exchanges["sparse_rows"] += m
        self._pending = ("sparse", (w1, w2, all_ids, all_rec), (ids, rec[:m]))
        return True

    def _finish_sparse(self, payload, local):
        w1, w2, all_ids, all_rec = payload
        ids_local, rec_local = local
        w1.wait()
        w2.wait()
        n, k = self.total_items, self.k
        valid = all_ids >= 0
        uniq, inv = torch.unique(all_ids[valid], return_inverse=True)
        if uniq.numel() == 0:
            return
        recs = all_rec[valid]
        S = torch.zeros((uniq.numel(), k + 1), dtype=torch.float32, device=self.flat.device).index_add_(0, inv, recs)
        wV, wB = self._weights(recs[:, :k], recs[:, k])
        cV = torch.zeros(uniq.numel(), dtype=torch.float32, device=self.flat.device).index_add_(0, inv, wV)
        cB = torch.zeros(uniq.numel(), dtype=torch.float32, device=self.flat.device).index_add_(0, inv, wB)
        RV = S[:, :k] * self._factors(S[:, :k], cV).unsqueeze(1)
        RB = S[:, k] * self._factors(S[:, k], cB)
        V, B = self.V, self.B
        baseV, baseB = self.base[: n * k].view(n, k), self.base[n * k:]
        # base' = base + R, flat' = base' + ((flat - base) - d_local) on the union of the touched rows (the dense form's
        # arithmetic, so an untrained row ends with flat' == base' bit for bit)
        dl = torch.zeros((uniq.numel(), k + 1), dtype=torch.float32, device=self.flat.device)
        if ids_local.numel():
            dl[torch.searchsorted(uniq, ids_local)] = rec_local
        bV, bB = baseV[uniq], baseB[uniq]
        pV, pB = (V[uniq] - bV).sub_(dl[:, :k]), (B[uniq] - bB).sub_(dl[:, k])
        bV.add_(RV)
        bB.add_(RB)
        baseV[uniq] = bV
        baseB[uniq] = bB
        V[uniq] = bV + pV
        B[uniq] = bB + pB

    # ---- resident protocol (ShardedBprTrainer.run_epoch_resident): several exchanges may be in flight --------------------
    # publish: d = flat - base (this rank's steps since the last publication), base = flat, all-reduce [d | weights];
    # apply (any time after the sum has landed): c = rule(S) - d_own, flat += c, base += c.  flat - base stays "steps not
    # yet published" throughout, whatever the number of exchanges in flight; with every exchange applied exactly one
    # boundary after its publication the tables are those of begin_sync / finish_sync.  On a GPU the bins' own workgroups
    # do both per row inside the epoch's launch (csrc/bpr_ldsbin.inc); this torch form is the gloo path and the
    # statement the device tests compare with.
    def resident_publish(self):
        n, k = self.total_items, self.k
        bucket = torch.empty(n * k + 3 * n, dtype=torch.float32, device=self.flat.device)
        delta = bucket[: n * k + n]
        torch.sub(self.flat, self.base, out=delta)
        self.base.copy_(self.flat)
        wV, wB = self._weights(delta[: n * k].view(n, k), delta[n * k:])
        bucket[n * k + n: n * k + 2 * n] = wV
        bucket[n * k + 2 * n:] = wB
        keep = delta.clone()
        work = None
        if dist.is_available() and dist.is_initialized():
            work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.exchanges["dense"] += 1
        return work, bucket, keep

    def resident_correction(self, bucket, keep):
        """c = rule(S) - keep for an all-reduced bucket [S | weights]"""
        n, k = self.total_items, self.k
        c = torch.empty(n * k + n, dtype=torch.float32, device=bucket.device)
        S = bucket[: n * k].view(n, k)
        torch.mul(S, self._factors(S, bucket[n * k + n: n * k + 2 * n]).unsqueeze(1), out=c[: n * k].view(n, k))
        torch.mul(bucket[n * k: n * k + n], self._factors(bucket[n * k: n * k + n], bucket[n * k + 2 * n:]), out=c[n * k:])
        return c.sub_(keep)

    def resident_apply(self, work, bucket, keep):
        if work is not None:
            work.wait()
        c = self.resident_correction(bucket, keep)
        self.flat.add_(c)
        self.base.add_(c)

    def step_sync(self):
        """finish_sync() of the pending exchange followed by begin_sync() of the next one; on a GPU the two table
        passes are one fused kernel"""
        if (self._pending is None or self._pending[0] is None or self._pending[0] == "sparse" or self.trainer is None
                or not self.flat.is_cuda or self.sparse_threshold is not None):
            self.finish_sync()
            self.begin_sync()
            return
        work, bucket_prev, local_prev = self._pending
        self._pending = None
        work.wait()
        self._await_emulated()
        n, k = self.total_items, self.k
        bucket, local = self._next_buffers()
        self.trainer.table_delta_step(self.flat.data_ptr(), self.base.data_ptr(), bucket_prev.data_ptr(),
                                      local_prev.data_ptr(), n, k, bucket.data_ptr(), local.data_ptr(), rule=RULES[self.rule])
        self.exchanges["dense"] += 1
        work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._emulate_collective(bucket)
        self._pending = (work, bucket, local)

    def finish_sync(self):
        if self._pending is None:
            return
        work, bucket, local = self._pending
        self._pending = None
        if work is None:
            self.base.copy_(self.flat)
            return
        if work == "sparse":
            self._finish_sparse(bucket, local)
            return
        work.wait()  # stream-level wait on CUDA, blocking on gloo
        self._await_emulated()
        n, k = self.total_items, self.k
        if self.trainer is not None and self.flat.is_cuda:
            self.trainer.table_delta_finish(self.flat.data_ptr(), self.base.data_ptr(), bucket.data_ptr(),
                                            local.data_ptr(), n, k, rule=RULES[self.rule])
            return
        delta = bucket[: n * k + n]
        dV = delta[: n * k].view(n, k)
        dV.mul_(self._factors(dV, bucket[n * k + n: n * k + 2 * n]).unsqueeze(1))
        delta[n * k:].mul_(self._factors(delta[n * k:], bucket[n * k + 2 * n:]))
        # base' = base + R, flat' = base' + ((flat - base) - local): a row nobody trained since begin_sync ends with
        # flat' == base' bit for bit (see table_delta_finish_kernel), so the next exchange sees it as untouched
        progress = (self.flat - self.base).sub_(local)
        self.base.add_(delta)
        torch.add(self.base, progress, out=self.flat)


class ShardedBprTrainer:
    """Drives one rank's cornac_hip BPR handle plus the replicated item table."""

    def __init__(self, trainer, total_items, k, device, sync_every, group=None, sparse_threshold=None, rule="sqrt"):
        self.trainer = trainer
        self.table = ItemTableReplica(total_items, k, device, group, trainer=trainer if device.type == "cuda" else None,
                                      sparse_threshold=sparse_threshold, rule=rule)
        self._epochs_since_exchange = 0
        self.sync_every = int(sync_every)
        self.device = device
        self.stream = None
        self._resident = None            # buffers of the resident exchange (device path), allocated at first use
        self._resident_open = False      # resident launches since the last closing exchange (finish())
        self._carry = (0, 0)             # counters fetched by a sync() outside finish() (load_items)
        self.resident_timeout_ms = 20000
        self.resident_lag = 1            # host path: an exchange is applied this many boundaries after its publication
        # emulation / test hook: called as hook(e, bucket) on the communication stream where the all-reduce of exchange e
        # sits (after it, when there is one) — e.g. `bucket *= 2` plays a twin rank on a single GPU
        self.resident_bucket_hook = None
        if device.type == "cuda":
            self.stream = torch.cuda.Stream(device)
            if trainer is not None:
                torch.cuda.synchronize(device)
                trainer.bind_device(None, self.table.V.data_ptr(), self.table.B.data_ptr())
                trainer.set_stream(self.stream.cuda_stream)
                # dense exchange: the replica is touched between chunks only by the handle's own table passes, so the XCD-strata
                # form may keep its packed item records (row + bias line in one place) from chunk to chunk; the passes then
                # work on the records and sync() writes them back.  The sparse exchange indexes the dense replica from torch.
                if sparse_threshold is None and hasattr(trainer, "chunk_records") and dist.is_available() and dist.is_initialized():
                    trainer.chunk_records(True)

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def load_items(self, V, B):
        with self._on_stream():
            if self.device.type == "cuda" and self.trainer is not None and hasattr(self.trainer, "sync"):
                # (chunk_records mode: the handle may be training on packed item records; a load into the dense replica would be
                # overwritten by the stale records at the next write-back — make the dense table current first; advisor r4)
                c, s = self.trainer.sync()
                self._carry = (self._carry[0] + c, self._carry[1] + s)
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

    def run_epoch_in_parts(self, nnz, parts, lr, reg, use_bias=True, neg_population=0, flags=0):
        """one epoch of `nnz` samples as EXACTLY `parts` chunks (sizes differ by at most one) — the form for ranks whose
        sample counts differ: every rank issues the same number of exchanges whatever its nnz (run() derives the chunk
        count from the sample count)"""
        nnz, parts = int(nnz), int(parts)
        with self._on_stream():
            for c in range(parts):
                n = nnz * (c + 1) // parts - nnz * c // parts
                if n:
                    self.trainer.hogwild_enqueue(n, lr, reg, use_bias, neg_population, flags)
                self.table.step_sync()

    # ---- resident exchange: ONE launch per epoch, the exchange points inside it (csrc/bpr_ldsbin.inc) ----------------------
    def resident_bins(self, neg_population=0, flags=0):
        """arrivals per exchange of the resident launch, 0 when this trainer / table cannot take it (not the LDS-bin form,
        sparse exchange asked for, a host stand-in without the entry point)"""
        if (self.table.sparse_threshold is not None or self.device.type != "cuda"
                or not hasattr(self.trainer, "resident_exchange_bins")):
            return 0
        return int(self.trainer.resident_exchange_bins(neg_population, flags))

    def _resident_buffers(self, n_ex):
        st = self._resident
        if st is None or st["n_ex"] != n_ex:
            n, k, dev = self.table.total_items, self.table.k, self.device
            st = self._resident = {
                "n_ex": n_ex,
                # every exchange of an epoch has its own bucket and keep buffer (n_ex x 2 x the table: 230 MB at the ML-20M
                # shape with 16 exchanges — nothing on a 288 GB part): the launch never has to wait for a buffer
                "buckets": torch.zeros((n_ex, n * k + 3 * n), dtype=torch.float32, device=dev),
                "keeps": torch.zeros((n_ex, n * k + n), dtype=torch.float32, device=dev),
                "signals": torch.zeros(2 * n_ex + 1, dtype=torch.int32, device=dev),   # arrive | landed | error
                "applied": torch.zeros(n, dtype=torch.int32, device=dev),
                "comm": torch.cuda.Stream(dev),
            }
            torch.cuda.synchronize(dev)
        return st

    def run_epoch_resident(self, n_exchanges, lr, reg, use_bias=True, neg_population=0, flags=0):
        """one epoch as ONE launch that publishes the item-table deltas at `n_exchanges` points and applies the landed sums
        itself; this side only feeds the collectives: on the communication stream, per exchange, wait for the launch's
        arrivals -> all-reduce the bucket in place -> raise the landed flag.  The launch never waits, so nothing here can
        deadlock it; a wait that does not end within resident_timeout_ms is reported by finish()."""
        from . import _lib

        t, n_ex = self.table, int(n_exchanges)
        st = self._resident_buffers(n_ex)
        sig, comm, dev_index = st["signals"], st["comm"], self.device.index or 0
        p_arrive, p_landed, p_err = sig.data_ptr(), sig.data_ptr() + 4 * n_ex, sig.data_ptr() + 8 * n_ex
        stride_b, stride_k = st["buckets"].stride(0), st["keeps"].stride(0)
        with self._on_stream():
            if t._pending is not None:
                t.finish_sync()            # (an exchange of the chunk protocol still in flight)
            sig[: 2 * n_ex].zero_()
            zeroed = torch.cuda.Event()
            zeroed.record()
            arrivals = self.trainer.epoch_resident_enqueue(lr, reg, use_bias, neg_population, flags, n_ex, RULES[t.rule],
                                                           t.base.data_ptr(), st["buckets"].data_ptr(), stride_b,
                                                           st["keeps"].data_ptr(), stride_k, p_arrive, p_landed,
                                                           st["applied"].data_ptr())
        comm.wait_event(zeroed)
        collective = dist.is_available() and dist.is_initialized()
        with torch.cuda.stream(comm):
            for e in range(n_ex):
                _lib.stream_wait_counter(dev_index, comm.cuda_stream, p_arrive + 4 * e, arrivals, p_err, self.resident_timeout_ms)
                if collective:
                    dist.all_reduce(st["buckets"][e], op=dist.ReduceOp.SUM, group=t.group, async_op=True).wait()
                t._emulate_collective(st["buckets"][e], stream=comm.cuda_stream)
                if self.resident_bucket_hook is not None:
                    self.resident_bucket_hook(e, st["buckets"][e])
                # a wait that gave up (p_err set) leaves this and every later flag of the epoch down: a bucket that was
                # all-reduced incomplete is applied neither by the launch nor by the flush; finish() raises on every rank
                _lib.stream_set_flag(dev_index, comm.cuda_stream, p_landed + 4 * e, 1, p_err)
            landed = torch.cuda.Event()
            landed.record()
        with self._on_stream():
            self.stream.wait_event(landed)
            self.trainer.resident_flush(n_ex, RULES[t.rule], t.base.data_ptr(), st["buckets"].data_ptr(), stride_b,
                                        st["keeps"].data_ptr(), stride_k, st["applied"].data_ptr(), p_landed)
        t.exchanges["dense"] += n_ex
        t.exchanges["resident"] = t.exchanges.get("resident", 0) + n_ex
        # the hot rows live in the global table: their owner bin publishes them when IT has finished, and bins that train
        # on add steps afterwards — table - base still holds those on this rank only.  finish() closes with one exchange.
        self._resident_open = True

    def _run_epoch_resident_host(self, nnz, n_exchanges, lr, reg, use_bias, neg_population, flags):
        """the same protocol on a host (gloo): the stand-in trains chunk by chunk, the table does for all rows at once
        what the device's duty waves do row by row"""
        t, pending = self.table, []
        if t._pending is not None:
            t.finish_sync()
        for e in range(n_exchanges):
            n = nnz * (e + 1) // n_exchanges - nnz * e // n_exchanges
            if n:
                self.trainer.hogwild_enqueue(n, lr, reg, use_bias, neg_population, flags)
            while pending and pending[0][0] <= e - self.resident_lag:
                t.resident_apply(*pending.pop(0)[1])
            ex = t.resident_publish()
            if self.resident_bucket_hook is not None:
                if ex[0] is not None:
                    ex[0].wait()
                self.resident_bucket_hook(e, ex[1])
                ex = (None, ex[1], ex[2])
            pending.append((e, ex))
        for _, ex in pending:              # (the device's flush)
            t.resident_apply(*ex)
        t.exchanges["resident"] = t.exchanges.get("resident", 0) + n_exchanges

    def run_epoch(self, nnz, parts, lr, reg, use_bias=True, neg_population=0, flags=0, resident=None, epochs_per_exchange=1):
        """one epoch of `nnz` samples with `parts` item-table exchanges.  resident = None: the resident exchange (one
        launch) where the trainer has it, else `parts` chunk launches with the overlapped exchange between them
        (run_epoch_in_parts); True / False force one or the other (True on a host stand-in: the torch form).
        epochs_per_exchange > 1 (sparse item sides, exchange_schedule): the epoch is enqueued whole and only every
        epochs_per_exchange-th call exchanges (overlapped with the epochs that follow; finish() lands the last one)."""
        parts = int(parts)
        if int(epochs_per_exchange) > 1:
            with self._on_stream():
                self.trainer.hogwild_enqueue(int(nnz), lr, reg, use_bias, neg_population, flags)
                self._epochs_since_exchange += 1
                if self._epochs_since_exchange >= int(epochs_per_exchange):
                    self.table.step_sync()
                    self._epochs_since_exchange = 0
            return
        if resident is None:
            resident = 1 <= parts <= 32 and self.resident_bins(neg_population, flags) > 0
        if not resident:
            return self.run_epoch_in_parts(nnz, parts, lr, reg, use_bias, neg_population, flags)
        if self.device.type == "cuda" and hasattr(self.trainer, "epoch_resident_enqueue"):
            return self.run_epoch_resident(parts, lr, reg, use_bias, neg_population, flags)
        return self._run_epoch_resident_host(int(nnz), parts, lr, reg, use_bias, neg_population, flags)

    def finish(self):
        with self._on_stream():
            if self._epochs_since_exchange:      # epochs trained since the last exchange of a multi-epoch schedule
                self.table.step_sync()
                self._epochs_since_exchange = 0
            self.table.finish_sync()
            if self._resident_open:
                # closing exchange of the resident protocol: the steps taken on the hot rows after their last in-launch
                # publication (table - base = steps not yet published, the invariant the chunk protocol's passes share)
                # reach the other ranks, so that every rank leaves with the same item table
                self.table.begin_sync()
                self.table.finish_sync()
                self._resident_open = False
        out = self.trainer.sync()
        if self._carry != (0, 0) and isinstance(out, tuple) and len(out) == 2:
            out, self._carry = (out[0] + self._carry[0], out[1] + self._carry[1]), (0, 0)
        if self.stream is not None:
            self.stream.synchronize()
        if self._resident is not None:
            self._resident["comm"].synchronize()
            err = self._resident["signals"][-1:].clone()
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(err, op=dist.ReduceOp.MAX, group=self.table.group)   # every rank raises together
            if int(err.item()) != 0:
                self._resident["signals"][-1:].zero_()
                # (the flag is per rank and sticky: the rank whose wait gave up withholds the landed flag of that exchange and of
                # every later one — of later epochs too — until this raise; the other ranks went on applying sums that lack
                # its contribution.  Fatal either way: the MAX over the ranks makes every rank raise here together.)
                raise RuntimeError("resident exchange: the communication stream of at least one rank gave up waiting for its "
                                   "epoch launch's arrivals (%d ms); that rank applied none of the sums from that exchange on, "
                                   "the others applied sums without its deltas: the item tables of the ranks no longer agree "
                                   "— restart the fit from the last consistent model" % self.resident_timeout_ms)
        return out


class ShardedMfTrainer:
    """Multi-GPU MF (fit_sgd, backend_cpu.pyx:35-97), regime 1: the ratings are partitioned BY USER, so every rank keeps
    its users' rows of U and Bu to itself (no collective for them); the item side [V | Bi] is replicated and reconciled
    like BPR's item table (ItemTableReplica: deltas summed over the ranks, divided by sqrt(touching ranks), sparse
    records when few rows moved).  An epoch of a rank = its own ratings once, enqueued in `parts_per_epoch` slices of the
    stored order; the exchange of slice p is in flight while slice p + 1 trains.  `mu` is the GLOBAL mean rating
    (global_mean_across_ranks).  The slices run the fused atomic kernel (the block rotation needs whole epochs:
    parts_per_epoch = 1 picks it where cornac_hip_mf_fit would).  The item side is reconciled with the "align" rule
    (ItemTableReplica); parts_per_epoch = None: 8 exchanges per epoch, 16 from 5 ranks on (see __init__)."""

    def __init__(self, trainer, total_items, k, device, parts_per_epoch=None, group=None, sparse_threshold=None,
                 rule="align"):
        self.trainer = trainer
        if parts_per_epoch is None:
            # Emulation of this algebra with the oracle's fit_sgd loop (tools/emulate_ranks_mf.py,
            # profiles/r04_emulate_ranks_mf.log).  Round 3's sum / sqrt(c) rule diverged at R = 8 below 16 exchanges per
            # epoch (MF's squared-error steps are not bounded like BPR's sigmoid) and needed 4 R of them; the "align" rule
            # (ItemTableReplica) cannot diverge that way — R copies of one step are averaged — and at R = 8 reaches the
            # held-out RMSE of ONE process on all ratings within 0.017 at 8 exchanges per epoch and 0.006 at 16 (0.445 /
            # 0.434 against 0.428; twice the ratings per user: 0.427 / 0.419 against 0.417).  Hence 8, and 16 from 5
            # ranks on.  With rule="sqrt" the round-3 count is kept.
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
            parts_per_epoch = (8 if world <= 4 else 16) if rule == "align" else max(16, 4 * world)
        self.table = ItemTableReplica(total_items, k, device, group, trainer=trainer if device.type == "cuda" else None,
                                      sparse_threshold=sparse_threshold, rule=rule)
        self.parts = max(1, int(parts_per_epoch))
        self.device = device
        self.stream = None
        if device.type == "cuda":
            self.stream = torch.cuda.Stream(device)
            if trainer is not None:
                torch.cuda.synchronize(device)
                trainer.bind_items(self.table.V.data_ptr(), self.table.B.data_ptr())
                trainer.set_stream(self.stream.cuda_stream)

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def load_items(self, V, Bi):
        with self._on_stream():
            self.table.load(V, Bi)
        if self.stream is not None:
            self.stream.synchronize()

    def run_epoch(self, lr, reg, mu, use_bias=True):
        with self._on_stream():
            for part in range(self.parts):
                self.trainer.epoch_enqueue(part, self.parts, lr, reg, mu, use_bias)
                self.table.step_sync()

    def finish(self):
        """completes the pending exchange; returns the sum of squared errors of this rank's ratings since the last
        finish() (0.5 x the all-reduced sum is the reference's epoch loss, backend_cpu.pyx:86-88)"""
        with self._on_stream():
            self.table.finish_sync()
        out = self.trainer.sync()
        if self.stream is not None:
            self.stream.synchronize()
        return out


def global_mean_across_ranks(values, group=None):
    """train_set.global_mean of the union of the ranks' ratings: one all-reduce of (sum, count) in float64"""
    t = torch.tensor([float(np.sum(values, dtype=np.float64)), float(len(values))], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t = t.cpu()
    return float(t[0] / max(float(t[1]), 1.0))


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
        self.mark = lambda label: None   # tracing hook (RowShardedBprTrainer.trace)

    def owner_major(self, item_ids):
        return (item_ids % self.world) * self.rows_per_rank + item_ids // self.world

    def load(self, V, B):
        """keeps this rank's rows of the full host tables"""
        mine = np.arange(self.rank, self.total_items, self.world)
        self.V[: len(mine)].copy_(torch.as_tensor(np.ascontiguousarray(np.asarray(V, np.float32)[mine])))
        self.B[: len(mine)].copy_(torch.as_tensor(np.ascontiguousarray(np.asarray(B, np.float32)[mine])))

    def _exchange(self, send, send_counts, recv_counts, group=None):
        if not self.collective:
            return send
        recv = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                               group=group if group is not None else self.group)
        return recv

    def dedupe_items(self, i, j):
        """De-duplicate the items of a micro-batch (int32 device arrays `i`, `j`; entries < 0 are skipped draws)
        WITHOUT sorting and with ONE host synchronisation: a flag per table row in owner-major order, an inclusive scan
        of the flags (slot of a touched row = scan - 1; the scan order is bucketed by owner), the per-owner counts read
        off the scan at the owner boundaries, exchanged with the peers on the device and copied to the host together
        with the local counts.  Returns (local_rows int32 [n_unique]: the request lists, one run per owner; slot_i,
        slot_j int32 [m]; send_counts; recv_counts).  The mark / slot / request-list passes are HIP kernels when `ops`
        has them (DeviceRowOps), device-agnostic torch code otherwise (the gloo tests)."""
        n_rows = self.world * self.rows_per_rank
        fast = hasattr(self.ops, "shard_mark")
        if fast:
            mark = torch.zeros(n_rows, dtype=torch.int32, device=i.device)
            self.ops.shard_mark(i, j, self.world, self.rows_per_rank, mark)
            scan = torch.cumsum(mark, 0, dtype=torch.int32)
            slot_i, slot_j = torch.empty_like(i), torch.empty_like(j)
            self.ops.shard_slots(i, j, self.world, self.rows_per_rank, scan, slot_i, slot_j)
        else:
            valid = i >= 0
            g = torch.cat([self.owner_major(i.long()), self.owner_major(j.long())])
            gg = torch.where(torch.cat([valid, valid]), g, torch.full_like(g, n_rows))   # skipped -> a spare flag
            mark = torch.zeros(n_rows + 1, dtype=torch.int32, device=i.device)
            mark[gg] = 1
            mark[n_rows] = 0
            scan = torch.cumsum(mark, 0, dtype=torch.int32)
            slot = (scan[gg] - 1).clamp_(min=0)
            slot_i, slot_j = slot[: len(i)].contiguous(), slot[len(i):].contiguous()
        self.mark("a slots")
        bounds = torch.arange(1, self.world + 1, device=i.device) * self.rows_per_rank - 1
        ends = scan[bounds].to(torch.int64)
        sc = ends - torch.cat([ends.new_zeros(1), ends[:-1]])
        if self.collective:
            rc = torch.empty_like(sc)
            dist.all_to_all_single(rc, sc, group=self.group)
        else:
            rc = sc
        self.mark("a counts")
        counts = torch.stack([sc, rc]).tolist()                           # the ONE host synchronisation
        self.mark("a synced")
        send_counts, recv_counts = counts[0], counts[1]
        n_unique = int(sum(send_counts))
        if fast:
            local_rows = torch.empty(n_unique, dtype=torch.int32, device=i.device)
            if n_unique:
                self.ops.shard_uniq(mark, scan, self.rows_per_rank, local_rows)
        else:
            local_rows = (torch.nonzero(mark[:n_rows]).view(-1) % self.rows_per_rank).to(torch.int32)
        return local_rows, slot_i, slot_j, send_counts, recv_counts

    def fetch(self, uniq_g, send_counts=None, recv_counts=None):
        """uniq_g: sorted unique owner-major indices (int64, on `device`).  Returns (rows [n,k], bias [n], plan).
        The split sizes are derived here with two host syncs (the training loop gets them from `dedupe_items`)."""
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
        return self.fetch_local((uniq_g % self.rows_per_rank).to(torch.int32), send_counts, recv_counts)

    def fetch_local(self, local_rows, send_counts, recv_counts):
        """local_rows: the request lists (`dedupe_items`).  The plan keeps the rows as the owner SENT them: a later
        `push_updated` turns the returned rows into deltas on the owner's side."""
        wanted = self._exchange(local_rows, send_counts, recv_counts)          # rows the others want from me
        self.mark("a ids")
        out_rows = torch.empty(len(wanted), self.k, dtype=torch.float32, device=self.device)
        out_bias = torch.empty(len(wanted), 1, dtype=torch.float32, device=self.device)
        self.ops.gather(self.V, wanted, out_rows)
        self.ops.gather(self.B.view(-1, 1), wanted, out_bias)
        self.mark("a gathered")
        rows = self._exchange(out_rows, recv_counts, send_counts)
        bias = self._exchange(out_bias, recv_counts, send_counts).view(-1)
        if not self.collective:   # no exchange: the staged rows must not alias what the owner keeps
            rows, bias = rows.clone(), bias.clone()
        return rows, bias, (send_counts, recv_counts, wanted, out_rows, out_bias)

    def _sender_scale(self, wanted):
        """1 / sqrt(number of ranks that requested the row), per received row (None on one rank)"""
        if self.world == 1 or not len(wanted):
            return None
        idx = wanted.long()
        senders = torch.zeros(self.rows_per_rank, dtype=torch.float32, device=self.device)
        senders.index_add_(0, idx, torch.ones(len(idx), dtype=torch.float32, device=self.device))
        return senders[idx].rsqrt()

    def push(self, plan, d_rows, d_bias):
        """owners apply  sum of the received deltas of a row / sqrt(number of ranks that sent one)  (the same
        reconciliation rule as ItemTableReplica: several ranks' stale steps on one popular row are damped)"""
        send_counts, recv_counts, wanted = plan[:3]
        got_rows = self._exchange(d_rows.contiguous(), send_counts, recv_counts)
        got_bias = self._exchange(d_bias.contiguous().view(-1, 1), send_counts, recv_counts)
        scale = self._sender_scale(wanted)
        if scale is not None:
            got_rows = got_rows * scale.unsqueeze(1)
            got_bias = got_bias * scale.unsqueeze(1)
        self.ops.scatter_add(self.V, wanted, got_rows)
        self.ops.scatter_add(self.B.view(-1, 1), wanted, got_bias)

    def push_updated(self, plan, rows, bias, group=None):
        """the same push from the UPDATED staged rows: they travel back as they are and the owner applies
        (returned - sent) / sqrt(senders) in one scatter pass — no delta buffer on the requester's side"""
        send_counts, recv_counts, wanted, out_rows, out_bias = plan
        got_rows = self._exchange(rows.contiguous(), send_counts, recv_counts, group)
        got_bias = self._exchange(bias.contiguous().view(-1, 1), send_counts, recv_counts, group)
        scale = self._sender_scale(wanted)
        if hasattr(self.ops, "scatter_diff"):
            self.ops.scatter_diff(self.V, wanted, got_rows, out_rows, scale)
            self.ops.scatter_diff(self.B.view(-1, 1), wanted, got_bias, out_bias, scale)
        else:
            d_rows, d_bias = got_rows - out_rows, got_bias - out_bias
            if scale is not None:
                d_rows, d_bias = d_rows * scale.unsqueeze(1), d_bias * scale.unsqueeze(1)
            self.ops.scatter_add(self.V, wanted, d_rows)
            self.ops.scatter_add(self.B.view(-1, 1), wanted, d_bias)
        return got_rows, got_bias

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
    """row gather / scatter and the de-duplication passes through libcornac_hip on the trainer's stream (device
    tensors only)"""

    def __init__(self, trainer):
        self.trainer = trainer

    def gather(self, table, ids, out):
        self.trainer.gather_rows(table.data_ptr(), ids.data_ptr(), len(ids), table.shape[1], out.data_ptr())

    def scatter_add(self, table, ids, delta):
        self.trainer.scatter_add_rows(table.data_ptr(), ids.data_ptr(), len(ids), table.shape[1], delta.data_ptr())

    def scatter_diff(self, table, ids, now, before, scale=None):
        self.trainer.scatter_diff_rows(table.data_ptr(), ids.data_ptr(), len(ids), table.shape[1], now.data_ptr(),
                                       before.data_ptr(), scale.data_ptr() if scale is not None else None)

    def shard_mark(self, i, j, world, rows_per_rank, mark):
        self.trainer.shard_mark(i.data_ptr(), j.data_ptr(), len(i), world, rows_per_rank, mark.data_ptr())

    def shard_slots(self, i, j, world, rows_per_rank, scan, slot_i, slot_j):
        self.trainer.shard_slots(i.data_ptr(), j.data_ptr(), len(i), world, rows_per_rank, scan.data_ptr(),
                                 slot_i.data_ptr(), slot_j.data_ptr())

    def shard_uniq(self, mark, scan, rows_per_rank, uniq_local):
        self.trainer.shard_uniq(mark.data_ptr(), scan.data_ptr(), len(mark), rows_per_rank, uniq_local.data_ptr())


class RowShardedBprTrainer:
    """One rank of BPR with the item table sharded by row.  Users (CSR slice, U rows) are rank-local as in
    regime 1; each micro-batch goes through two stages:

        A  draw the triplets, de-duplicate their items, fetch the touched rows from their owners into a staging table
        B  the hogwild update on the staged rows, the updated rows back to their owners (who apply the difference)

    On a GPU the fetch side, the update kernel and the push side run on three streams (two RCCL communicators, one per
    exchanging stream): stage A of micro-batch r overlaps the update of r-1 and the push of r-2 (the update kernel
    waits on atomics and leaves the copy / exchange bandwidth free), so a fetch may miss the pushes of the three
    previous micro-batches — the usual bounded-staleness asynchrony.  With the handle's
    owned kernel (k in 33..256) stage A is its EMIT launch and stage B its STAGED launch: user rows stay with one wave
    and are updated with plain stores, as in the single-GPU kernel; otherwise the generic sample / apply kernels run."""

    BIAS_STRIDE = 32  # staged biases sit one per 128-byte line (see DESIGN.md "padded bias table")

    def __init__(self, trainer, total_items, k, device, micro_batch, group=None, ops=None, pipeline=None):
        self.trainer, self.device, self.micro_batch, self.group = trainer, device, int(micro_batch), group
        self.stream = self.stream_b = self.stream_c = None
        cuda = device.type == "cuda"
        self.pipeline = cuda if pipeline is None else bool(pipeline)
        if cuda:
            self.stream = torch.cuda.Stream(device)
            self.stream_b = torch.cuda.Stream(device) if self.pipeline else self.stream
            self.stream_c = torch.cuda.Stream(device) if self.pipeline else self.stream
            torch.cuda.synchronize(device)
            trainer.set_stream(self.stream.cuda_stream)
        with self._on(self.stream):
            self.table = RowShardedItemTable(total_items, k, device, ops or DeviceRowOps(trainer), group)
        # stage B's exchanges get their own communicator: two streams must not share one
        self.group_b = group
        if self.pipeline and self.table.collective:
            self.group_b = dist.new_group(ranks=None if group is None else dist.get_process_group_ranks(group))
        if cuda:
            # the handle's own (full-size) item table is not used in this mode: release it
            trainer.bind_device(None, self.table.V.data_ptr(), self.table.B.data_ptr())
        self.slots_cap = trainer.staged_slots(self.micro_batch) if cuda and hasattr(trainer, "staged_slots") else 0
        self.owned = self.slots_cap > 0
        self._trip, self._trip_free = [], []   # two sets of triplet arrays: stage A of r+1 writes while stage B of r reads
        self._turn, self._last_turn = 0, None
        self.rows_fetched = 0
        self._valid_draws = torch.zeros((), dtype=torch.int64, device=device)
        self.trace = None   # set to a list to collect (label, stream name, event) marks (tools/bench_sharded.py --trace)
        self.table.mark = lambda label: self._mark(label, self.stream, "A")

    def _mark(self, label, stream, which):
        if self.trace is not None and stream is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self.trace.append((label, which, ev))

    def _on(self, stream):
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def _on_stream(self):
        return self._on(self.stream)

    @property
    def triplets(self):
        """valid (not skipped) draws applied so far"""
        # the counter is accumulated on the driver's side stream: order the read after what is queued there
        if self.stream is not None:
            self.stream.synchronize()
        return int(self._valid_draws.item())

    def load_items(self, V, B):
        with self._on_stream():
            self.table.load(V, B)
        if self.stream is not None:
            self.stream.synchronize()

    def _sample(self, n):
        """(u, i, j) int32 device arrays; skipped draws (and padding slots) carry i = j = -1 and u < 0"""
        if self.owned:
            if not self._trip:
                for _ in range(2 if self.stream_b is not self.stream else 1):
                    self._trip.append(tuple(torch.empty(self.slots_cap, dtype=torch.int32, device=self.device)
                                            for _ in range(3)))
                self._trip_free = [None] * len(self._trip)
            turn = self._turn % len(self._trip)
            self._turn += 1
            u, i, j = self._trip[turn]
            if self._trip_free[turn] is not None:   # the stage B that read this set two micro-batches ago
                self.stream.wait_event(self._trip_free[turn])
            n_slots = self.trainer.emit_triplets(n, u.data_ptr(), i.data_ptr(), j.data_ptr(), self.slots_cap)
            self._last_turn = turn
            return u[:n_slots], i[:n_slots], j[:n_slots]
        u = torch.empty(n, dtype=torch.int32, device=self.device)
        i, j = torch.empty_like(u), torch.empty_like(u)
        self.trainer.sample_triplets(n, u.data_ptr(), i.data_ptr(), j.data_ptr())
        return u, i, j

    def _apply(self, u, slot_i, slot_j, rows, bias_pad, lr, reg, use_bias):
        fn = self.trainer.apply_staged if self.owned else self.trainer.apply_triplets
        fn(u.data_ptr(), slot_i.data_ptr(), slot_j.data_ptr(), len(u), rows.data_ptr(), bias_pad.data_ptr(),
           self.BIAS_STRIDE, lr, reg, use_bias)

    def _stage_a(self, n):
        """draw, de-duplicate, fetch (stream A)"""
        if self.stream is not None:
            self.trainer.switch_stream(self.stream.cuda_stream)
        with self._on(self.stream):
            self._mark("A begin", self.stream, "A")
            if n > 0:
                u, i, j = self._sample(n)
            else:   # nothing left to draw here: an empty batch keeps the collectives matched
                u = torch.full((0,), -1, dtype=torch.int32, device=self.device)
                i, j = u.clone(), u.clone()
            self._mark("A emitted", self.stream, "A")
            local_rows, slot_i, slot_j, send_counts, recv_counts = self.table.dedupe_items(i, j)
            self._mark("A deduped", self.stream, "A")
            rows, bias, plan = self.table.fetch_local(local_rows, send_counts, recv_counts)
            bias_pad = torch.zeros(len(bias), self.BIAS_STRIDE, dtype=torch.float32, device=self.device)
            bias_pad[:, 0] = bias
            self.rows_fetched += len(local_rows)
            self._valid_draws = self._valid_draws + (i >= 0).sum()   # stays on the device: read through .triplets
            self._mark("A end", self.stream, "A")
            ready = None
            if self.stream is not None and self.stream_b is not self.stream:
                ready = torch.cuda.Event()
                ready.record(self.stream)
        return dict(u=u, slot_i=slot_i, slot_j=slot_j, rows=rows, bias_pad=bias_pad, plan=plan, ready=ready,
                    n_unique=len(local_rows), turn=self._last_turn if self.owned and n > 0 else None)

    def _stage_b(self, st, lr, reg, use_bias):
        """the update on the staged rows (stream B), then the updated rows back to their owners (stream C: the push of
        micro-batch r runs beside the update of r+1)"""
        split = st["ready"] is not None
        if self.stream_b is not None:
            self.trainer.switch_stream(self.stream_b.cuda_stream)
        with self._on(self.stream_b):
            if split:
                self.stream_b.wait_event(st["ready"])
                for t in (st["u"], st["slot_i"], st["slot_j"], st["rows"], st["bias_pad"]):
                    t.record_stream(self.stream_b)   # allocated on stream A, used here
            self._mark("B begin", self.stream_b, "B")
            if len(st["u"]) and st["n_unique"]:
                self._apply(st["u"], st["slot_i"], st["slot_j"], st["rows"], st["bias_pad"], lr, reg, use_bias)
            self._mark("B end", self.stream_b, "B")
            if split:
                applied = torch.cuda.Event()
                applied.record(self.stream_b)
                if st["turn"] is not None:
                    self._trip_free[st["turn"]] = applied
        if self.stream_c is not None:
            self.trainer.switch_stream(self.stream_c.cuda_stream)
        with self._on(self.stream_c):
            if split:
                self.stream_c.wait_event(applied)
                for t in (st["rows"], st["bias_pad"], st["plan"][2], st["plan"][3], st["plan"][4]):
                    t.record_stream(self.stream_c)
            self._mark("C begin", self.stream_c, "C")
            self.table.push_updated(st["plan"], st["rows"], st["bias_pad"][:, 0], group=self.group_b)
            self._mark("C end", self.stream_c, "C")

    def run(self, n_samples, lr, reg, use_bias=True):
        """n_samples draws on this rank, in micro-batches.  Every micro-batch is a collective (all-to-alls in both
        stages), so all ranks must go through the same number of them: the ranks agree on the maximum up front and a
        rank whose user shard runs out of draws keeps serving the others' fetches / pushes with empty batches."""
        left = int(n_samples)
        rounds = (left + self.micro_batch - 1) // self.micro_batch
        if self.table.collective and self.table.world > 1:
            r = torch.tensor([rounds], dtype=torch.int64, device=self.device)
            with self._on_stream():
                dist.all_reduce(r, op=dist.ReduceOp.MAX, group=self.group)
            rounds = int(r.item())
        prev = None
        for _ in range(rounds):
            n = min(left, self.micro_batch)
            left -= n
            if self.pipeline:
                if prev is not None:
                    self._stage_b(prev, lr, reg, use_bias)       # enqueued first: overlaps the next stage A
                prev = self._stage_a(n)
            else:
                self._stage_b(self._stage_a(n), lr, reg, use_bias)
        if prev is not None:
            self._stage_b(prev, lr, reg, use_bias)
        if self.stream is not None and self.stream_b is not self.stream:
            self.stream.wait_stream(self.stream_b)               # a following run / finish sees every update and push
            self.stream.wait_stream(self.stream_c)
            self.trainer.set_stream(self.stream.cuda_stream)

    def finish(self):
        for st in (self.stream_b, self.stream_c):
            if st is not None:
                st.synchronize()
        out = self.trainer.sync()
        if self.stream is not None:
            self.stream.synchronize()
        return out


# ---- regime 2: a ring conveyor of item blocks ------------------------------------------------------------------------------
class _OnceWork:
    """a point-to-point work handle that is waited for at most once (gloo's send / receive works block for ever on a second
    wait: they wait for the NEXT completion of their buffer)"""

    def __init__(self, work):
        self.work, self.done = work, False

    def wait(self):
        if not self.done:
            self.work.wait()
            self.done = True


def ring_strides(world, rings):
    """the strides of up to `rings` conveyor rings over `world` ranks: rank r hands its block to rank r - s.  A stride must be
    coprime to the world size (the block has to visit every rank); s and world - s use the two directions of the same
    links, two different strides different links — N = 8: 1, 7, 3, 5 (four directed link sets)"""
    import math

    if world <= 1:
        return [1] * max(1, int(rings))  # (one rank: K groups of two blocks each — the launch granularity of K rings)
    good = [s for s in range(1, max(world, 2)) if math.gcd(s, world) == 1] or [1]
    order = []
    for s in good:                      # 1, N - 1, 3, N - 3, ...: fill both directions of a link before taking the next one
        for c in (s, world - s):
            if c in good and c not in order:
                order.append(c)
    return order[: max(1, int(rings))]


# ---- regime 2, current form: the conveyor's blocks are bin ranges of the epoch's deal ---------------------------------------
class _DeviceConveyorTrainer:
    """the rank's ONE cornac_hip BPR handle in conveyor layout (cornac_hip_bpr_conveyor_*): its CSR / CSC over the global item
    ids stay put; what changes per step is which block buffers the launch reads its bins' rows from"""

    def __init__(self, indptr, indices, n_users, n_items, k, U, stream, device_index):
        from . import _lib

        self.tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k, device=device_index)
        self.U, self.stream, self.device = U, stream, U.device
        self.tr.bind_device(self.U.data_ptr(), None, None)
        self.tr.set_stream(stream.cuda_stream)

    def seed_hogwild(self, seed):
        self.tr.seed_hogwild(seed)

    def conveyor_setup(self, n_blocks, rank_item, deal_seed):
        self.n_bins, self.bpb, self.cap = self.tr.conveyor_setup(n_blocks, rank_item, deal_seed, release_item_tables=True)
        return self.n_bins, self.bpb, self.cap

    def conveyor_layout(self, layout_epoch):
        """(slot_item [n_bins cap], item_slot [n_items]) int32 device tensors, written on the trainer's stream"""
        n_items = self.tr.n_items
        slot_item = torch.empty(self.n_bins * self.cap, dtype=torch.int32, device=self.device)
        item_slot = torch.empty(n_items, dtype=torch.int32, device=self.device)
        self.tr.conveyor_layout(layout_epoch, slot_item.data_ptr(), item_slot.data_ptr())
        return slot_item, item_slot

    def conveyor_enqueue(self, epoch, layout_epoch, blocks, bufs, lr, reg, use_bias, neg_population, flags):
        self.tr.conveyor_enqueue(epoch, layout_epoch, blocks, [b.data_ptr() for b in bufs], lr, reg, use_bias, neg_population, flags)

    def sync(self):
        return self.tr.sync()

    def close(self):
        self.tr.close()


class BinConveyorBprTrainer:
    """Multi-GPU BPR with the item table SHARDED BY ROW and never replicated (SURVEY.md 8e regime 2; BASELINE configs[4]:
    "item table row-sharded across 8 x MI355X via RCCL / xGMI"): DSGD's block rotation (Gemulla et al. 2011) laid on the
    xGMI ring, with the blocks cut from the LDS-bin deal of the epoch instead of from the item ids.

      * users are partitioned over the N ranks (their rows of U never move).  Every rank holds ONE handle over its users'
        interactions in global item ids, in the conveyor layout of the LDS-bin form (csrc/bpr.hip
        cornac_hip_bpr_conveyor_setup): the epoch's deal (csrc/bpr_ldsbin.inc: popularity strata, a keyed bijection inside
        each, one item of every group per bin; the key shared by all ranks) puts every item into one of n_bins bins, and
        conveyor block B is the bins [B bpb, (B + 1) bpb).  A block's buffer holds the rows of its bins in (bin, slot) order;
      * an epoch is 2 N steps.  In step t rank r trains ring block (2 p + t) % 2N (p its position on the ring): one
        launch over the block's bins — the draws of ITS users' interactions with the bins' items, the negative among the
        items sharing the positive's bin (the single-GPU LDS-bin sampler, bench.py config.sampling) — reading and writing
        the rows in the buffer.  Sum over the steps = its nnz draws per epoch;
      * a block trained in step t travels to rank r - s during step t + 1 and is trained there in step t + 2: every row is
        in exactly ONE place — in training or in flight — so there is no replica, no staleness, no reconciliation rule;
      * THE DEAL CHANGES WITH THE EPOCH (every `redeal_every` epochs): at the boundary every rank holds its home blocks and the
        rows move from their slot under the old key to their slot under the new one — one all_to_all_single over the ranks
        (table / N per rank, ~two steps' worth of link traffic).  So the blocks — like the bins — are re-dealt: any two items
        share a bin with probability ~1 / n_bins per epoch whatever their ids (tests/test_ldsbin_deal_cpu.py), where round 5's
        static residue classes i % 2NK excluded 1 - 1 / 2NK of all (positive, negative) pairs for the whole fit.  The
        reference draws the negative over ALL items (recom_bpr.pyx:235-238);
      * rings = K > 1: xGMI is a mesh of point-to-point links and one ring uses one of a GPU's seven.  The blocks are then
        K groups of 2 N (block b K + g is ring block b of ring g, stride s_g: ring_strides); a step trains one block of
        every ring in ONE launch over K bin ranges and moves K blocks over K links.

    The draws of a step on different ranks touch disjoint user rows AND disjoint item rows, so the parallel run equals the
    serial execution of the same steps and re-deals in any order (tests/test_dist_cpu.py runs both).  The reference has no
    counterpart (one process; the update it distributes is recom_bpr.pyx:252-265).

    No hot-item path in this layout: a very popular item makes its bin heavy (the bins of a launch are scheduled dynamically;
    the heaviest one bounds the step).  WBPR's popularity-weighted negative is drawn among the interactions of the RANK'S OWN
    users with the bin's items (local popularity).

    trainer_factory(indptr, indices, n_users, n_items, k, U): host stand-ins (gloo tests) with seed_hogwild / conveyor_setup /
    conveyor_layout / conveyor_enqueue / sync / close.  emulate_traffic (one rank only): the block that would travel is copied
    to the free buffer on the communication stream, so a one-GPU run carries the conveyor's memory traffic and dependencies."""

    def __init__(self, indptr, indices, n_users, n_items, k, device, group=None, trainer_factory=None, seed=0, deal_seed=None,
                 emulate_traffic=False, rings=1, n_train_items=None, item_order=None, redeal_every=1, virtual_world=None):
        """n_items: rows of the item table (the model's total_items); n_train_items (default: all): the items training may
        touch — ids below it (the reference draws positives and negatives among the train items only).  item_order: the
        popularity order of the train items all ranks share (None: this rank's own).  deal_seed: the SAME on every rank
        (default: derived from `seed`, which the ranks of fit_bpr_ring share); the draws are keyed by (seed, rank).
        virtual_world (one rank only): lay the conveyor out for that many ranks — 2 virtual_world blocks per ring, all of them
        this rank's — for one-GPU measurements of an N-rank step's launch size."""
        self.device, self.group, self.k = device, group, int(k)
        self.world, self.rank = _world(group)
        self.lay_world = int(virtual_world) if (virtual_world and self.world == 1) else self.world
        self.strides = ring_strides(self.lay_world, rings)
        self.K = len(self.strides)
        self.nb = 2 * self.lay_world               # blocks per ring = steps per epoch
        self.nb_total = self.nb * self.K
        self.n_items, self.n_users = int(n_items), int(n_users)
        self.n_train = self.n_items if n_train_items is None else int(n_train_items)
        self.redeal_every = max(1, int(redeal_every))
        cuda = device.type == "cuda"
        self.stream = torch.cuda.Stream(device) if cuda else None
        self.comm = torch.cuda.Stream(device) if cuda else None
        self.emulate_traffic = bool(emulate_traffic) and self.world == 1
        self.U = torch.zeros((self.n_users, self.k), dtype=torch.float32, device=device)
        indices = np.ascontiguousarray(indices, np.int32)
        self.nnz = len(indices)
        if trainer_factory is not None:
            self.trainer = trainer_factory(indptr, indices, self.n_users, self.n_train, self.k, self.U)
        else:
            self.trainer = _DeviceConveyorTrainer(indptr, indices, self.n_users, self.n_train, self.k, self.U, self.stream,
                                                  device.index or 0)
        self.trainer.seed_hogwild((int(seed) * 0x9E3779B97F4A7C15 + 7919 * self.rank + 1) & 0xFFFFFFFFFFFFFFFF)
        self.deal_seed = (int(seed) ^ 0xD1B54A32D192ED03 if deal_seed is None else int(deal_seed)) & 0xFFFFFFFFFFFFFFFF
        self.n_bins, self.bpb, self.cap = self.trainer.conveyor_setup(self.nb_total, item_order, self.deal_seed)
        if self.n_bins != self.bpb * self.nb_total:
            raise RuntimeError("conveyor layout: %d bins are not %d blocks of %d" % (self.n_bins, self.nb_total, self.bpb))
        self.W = self.bpb * self.cap              # slots (rows) per block
        # position of this rank on ring g: the rank it hands to (rank - s_g) has position - 1
        self.pos = [(self.rank * pow(s, -1, self.world)) % self.world if self.world > 1 else 0 for s in self.strides]
        width = self.W * (self.k + 1)
        n_home = self.nb if self.world == 1 else 2   # (one rank: every block of the ring is its own)
        self.bufs = [[torch.zeros(width, dtype=torch.float32, device=device) for _ in range(n_home + 1)] for _ in range(self.K)]
        self.where = [self._home_where(p) for p in self.pos]        # per ring: ring block -> buffer
        self.arrived = [[None] * (n_home + 1) for _ in range(self.K)]  # per ring and buffer: event / works of the receive that fills it
        self._sent = []
        self.t = 0
        self.layout_epoch = 0
        self.steps_trained = []                    # (epoch step, global block): inspection / tests
        self.redeals = 0
        self._tail = None                          # rows [n_train, n_items) of the table: never trained, kept on the host
        self.timing = False                        # True (device only): events around every launch, transfer and re-deal
        self._events = {"launch": [], "move": [], "redeal": []}

    # ---- layout ----
    def _home_where(self, p):
        if self.world == 1:
            return {b: b for b in range(self.nb)}
        return {2 * p: 0, 2 * p + 1: 1}

    def block_id(self, g, b):
        """global block (= bin range) of ring g's block b"""
        return b * self.K + g

    def home_blocks(self, rank=None):
        """[(ring, ring block, global block)] a rank holds at an epoch boundary, in ascending global block order"""
        if self.world == 1:
            out = [(g, b, self.block_id(g, b)) for g in range(self.K) for b in range(self.nb)]
        else:
            r = self.rank if rank is None else rank
            out = []
            for g, s in enumerate(self.strides):
                p = (r * pow(s, -1, self.world)) % self.world
                out += [(g, 2 * p + h, self.block_id(g, 2 * p + h)) for h in (0, 1)]
        return sorted(out, key=lambda x: x[2])

    def _block_owner(self):
        """[nb_total] -> the rank a block is at home on"""
        own = np.zeros(self.nb_total, np.int64)
        for r in range(self.world):
            for _, _, B in self.home_blocks(r):
                own[B] = r
        return own

    def _views(self, g, buf):
        """(V [W, k], B [W]) of ring g's buffer `buf`"""
        flat = self.bufs[g][buf]
        return flat[: self.W * self.k].view(self.W, self.k), flat[self.W * self.k:]

    def _on(self, stream):
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def _timed(self, what, stream):
        """context: HIP events on `stream` around the block when self.timing (timing_summary() reads them)"""
        ring = self

        class _T:
            def __enter__(self_):
                self_.on = ring.timing and stream is not None
                if self_.on:
                    self_.a = torch.cuda.Event(enable_timing=True)
                    self_.a.record(stream)

            def __exit__(self_, *exc):
                if self_.on:
                    b = torch.cuda.Event(enable_timing=True)
                    b.record(stream)
                    ring._events[what].append((self_.a, b))

        return _T()

    def timing_summary(self):
        """{what: (count, mean ms, min ms)} of the launches (compute stream), the block transfers (communication stream) and the re-deals
        recorded since the last call; synchronises"""
        self._drain()
        out = {}
        for what, evs in self._events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[what] = (len(ms), float(np.mean(ms)) if ms else 0.0, float(np.min(ms)) if ms else 0.0)
            evs.clear()
        return out

    def _layout(self, layout_epoch):
        slot_item, item_slot = self.trainer.conveyor_layout(layout_epoch)
        return torch.as_tensor(slot_item).to(self.device), torch.as_tensor(item_slot).to(self.device)

    def load_items(self, V, B):
        """this rank's home blocks of every ring, from the full host tables, in the layout of epoch 0"""
        V, B = np.asarray(V, np.float32), np.asarray(B, np.float32)
        self._tail = (V[self.n_train:].copy(), B[self.n_train:].copy())
        self.where = [self._home_where(p) for p in self.pos]
        self.arrived = [[None] * len(self.bufs[0]) for _ in range(self.K)]
        self.t, self.layout_epoch = 0, 0
        with self._on(self.stream):
            slot_item, _ = self._layout(0)
            slot_item = slot_item.cpu().numpy()
            for g, b, blk in self.home_blocks():
                items = slot_item[blk * self.W: (blk + 1) * self.W]
                ok = items >= 0
                rows = np.zeros((self.W, self.k), np.float32)
                bias = np.zeros(self.W, np.float32)
                rows[ok], bias[ok] = V[items[ok]], B[items[ok]]
                v, bb = self._views(g, self.where[g][b])
                v.copy_(torch.as_tensor(rows))
                bb.copy_(torch.as_tensor(bias))
        if self.stream is not None:
            self.stream.synchronize()

    def set_user_factors(self, U):
        self.U.copy_(torch.as_tensor(np.ascontiguousarray(np.asarray(U, np.float32))))
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def get_user_factors(self):
        if self.stream is not None:
            self.stream.synchronize()
        return self.U.cpu().numpy()

    def _await(self, got):
        if got is None:
            return
        if self.stream is not None:
            self.stream.wait_event(got)
        else:
            for w in got:
                w.wait()

    # ---- the epoch-boundary re-deal ----
    def _owner_major(self):
        """the blocks in owner-major order: (om [nb_total] block -> its index in that order, om_inv, owner [nb_total] block ->
        rank, H = home blocks per rank); a rank's home slots are then the contiguous range [rank H W, (rank + 1) H W)"""
        if getattr(self, "_om", None) is None:
            H = len(self.home_blocks(0))
            om = np.zeros(self.nb_total, np.int64)
            owner = np.zeros(self.nb_total, np.int64)
            for r in range(self.world):
                for h, (_, _, blk) in enumerate(self.home_blocks(r)):
                    om[blk], owner[blk] = r * H + h, r
            om_inv = np.argsort(om)
            self._om = tuple(torch.as_tensor(x, device=self.device) for x in (om, om_inv, owner)) + (H,)
        return self._om

    def _fetch_layouts(self, old_epoch, new_epoch):
        """(slot -> item of the old deal, item -> slot of the new one, event): the layout kernels run on the TRAINING stream (the
        handle's), so they are enqueued in front of the launch the plan is to run beside"""
        with self._on(self.stream):
            old_slot_item, _ = self._layout(old_epoch)
            _, new_item_slot = self._layout(new_epoch)
            ev = None
            if self.stream is not None:
                ev = torch.cuda.Event()
                ev.record(self.stream)
        return old_slot_item, new_item_slot, ev

    def _plan_redeal(self, old_epoch, new_epoch, layouts=None):
        """the index side of a re-deal — where each of my rows goes in the send buffer, how many go to / come from every rank,
        which received row lands in which of my new slots: a pure function of the two layouts, which every rank has.  On the
        device it runs on its own stream, so that the plan of the NEXT boundary is computed beside the current epoch's last
        launch (step()) and the boundary itself only moves rows."""
        W, N = self.W, self.world
        plan_stream = getattr(self, "_plan_stream", None)
        if plan_stream is None and self.stream is not None:
            plan_stream = self._plan_stream = torch.cuda.Stream(self.device)
        old_slot_item, new_item_slot, ev = layouts if layouts is not None else self._fetch_layouts(old_epoch, new_epoch)
        with self._on(plan_stream):
            if ev is not None:
                plan_stream.wait_event(ev)
                old_slot_item.record_stream(plan_stream)
                new_item_slot.record_stream(plan_stream)
            om, om_inv, owner, H = self._owner_major()
            HW, me = H * W, self.rank
            if getattr(self, "_om_slots", None) is None:
                self._om_slots = (om_inv[:, None] * W + torch.arange(W, device=self.device)[None, :]).reshape(-1)
            # D[g]: the rank the row at old owner-major position g goes to (-1: an empty slot), T[g]: its new owner-major position
            items = old_slot_item[self._om_slots].long()
            valid = items >= 0
            nu = new_item_slot[items.clamp_(min=0)].long()
            T = om[nu // W] * W + nu % W
            D = torch.where(valid, owner[nu // W], torch.full_like(nu, -1))
            # sender: my rows, destination-major, inside a destination in the order of my old slots
            D_me = D[me * HW: (me + 1) * HW]
            onehot = D_me[None, :] == torch.arange(N, device=self.device)[:, None]
            run = onehot.cumsum(1)
            send_counts = run[:, -1]
            send_base = torch.cumsum(send_counts, 0) - send_counts
            dst = D_me.clamp(min=0)
            pos = torch.where(D_me >= 0, send_base[dst] + run.gather(0, dst[None, :])[0] - 1, torch.full_like(D_me, HW))
            # receiver: the rows for me, source-major, inside a source in the order of ITS old slots = a running count over D == me
            mine = D == me
            got = torch.cumsum(mine, 0)
            recv_counts = torch.stack([got[(r + 1) * HW - 1] for r in range(N)])
            recv_counts = recv_counts - torch.cat([recv_counts.new_zeros(1), recv_counts[:-1]])
            src_of = torch.full((HW,), HW, dtype=torch.long, device=self.device)        # my new local slot -> row of `recv` (HW: none)
            at = mine.nonzero().squeeze(1)
            src_of[T[at] - me * HW] = got[at] - 1
            rc = sc = None
            if N > 1:
                rc, sc = [int(c) for c in recv_counts.cpu()], [int(c) for c in send_counts.cpu()]
            done = None
            if plan_stream is not None:
                done = torch.cuda.Event()
                done.record(plan_stream)
        return {"key": (int(old_epoch), int(new_epoch)), "pos": pos, "src_of": src_of, "rc": rc, "sc": sc, "HW": HW, "done": done}

    def _redeal(self, new_layout_epoch):
        """every row from its slot under the deal of self.layout_epoch to its slot under the deal of new_layout_epoch; at an
        epoch boundary (every rank holds its home blocks).  One all_to_all_single of [row | bias] records: the message of rank r
        for rank s holds r's rows in the order of their OLD slots (owner-major), so the sender places a row by a running count
        per destination and the receiver finds it by a running count over the source's old slots — prefix sums over the slot
        tables, no sort; both sides compute them from the two layouts, which every rank has (_plan_redeal, usually already
        computed beside the previous epoch's last steps)."""
        W, k, N = self.W, self.k, self.world
        home = self.home_blocks()
        for g in range(self.K):                       # the last step's receives fill home buffers
            for buf in range(len(self.bufs[g])):
                self._await(self.arrived[g][buf])
                self.arrived[g][buf] = None
        for w in self._sent:
            w.wait()
        self._sent = []
        plan = getattr(self, "_plan", None)
        if plan is None or plan["key"] != (int(self.layout_epoch), int(new_layout_epoch)):
            plan = self._plan_redeal(self.layout_epoch, new_layout_epoch)
        self._plan = None
        with self._on(self.stream), self._timed("redeal", self.stream):
            if plan["done"] is not None:
                self.stream.wait_event(plan["done"])
            HW, pos, src_of = plan["HW"], plan["pos"], plan["src_of"]
            if self.stream is not None:               # (allocated on the plan's stream, read by this one)
                pos.record_stream(self.stream)
                src_of.record_stream(self.stream)
            if getattr(self, "_stage", None) is None:
                self._stage = [torch.zeros((HW + 1, k + 1), dtype=torch.float32, device=self.device) for _ in range(2 if N > 1 else 1)]
            send, recv = self._stage[0], self._stage[-1]
            for h, (g, b, blk) in enumerate(home):
                v, bb = self._views(g, self.where[g][b])
                send[:, :k].index_copy_(0, pos[h * W: (h + 1) * W], v)
                send[:, k].index_copy_(0, pos[h * W: (h + 1) * W], bb)
            if N > 1:
                rc, sc = plan["rc"], plan["sc"]
                dist.all_to_all_single(recv[: sum(rc)], send[: sum(sc)], rc, sc, group=self.group)
            recv[HW].zero_()
            for h, (g, b, blk) in enumerate(home):
                v, bb = self._views(g, self.where[g][b])
                idx = src_of[h * W: (h + 1) * W]
                torch.index_select(recv[:, :k], 0, idx, out=v)
                torch.index_select(recv[:, k], 0, idx, out=bb)
        self.layout_epoch = new_layout_epoch
        self.redeals += 1

    # ---- a step ----
    def step(self, lr, reg, use_bias=True, neg_population=0, flags=0):
        """one step of the conveyor: ONE launch trains, on every ring, the block whose turn it is; then the trained blocks go to
        the rings' next ranks and the ones their previous ranks have just trained arrive (beside the NEXT step's launch)"""
        epoch, ts = divmod(self.t, self.nb)
        if ts == 0:
            want = epoch - epoch % self.redeal_every
            if want != self.layout_epoch:
                self._redeal(want)
        ranks = dist.get_process_group_ranks(self.group) if (self.group is not None and self.world > 1) else list(range(self.world))
        moves = []                                    # (ring, trained block, its buffer, free buffer, block arriving)
        with self._on(self.stream):
            blocks, bufs = [], []
            for g, p in enumerate(self.pos):
                b = (2 * p + ts) % self.nb
                buf = self.where[g][b]
                self._await(self.arrived[g][buf])     # the receive that brought the block here
                self.arrived[g][buf] = None
                blocks.append(self.block_id(g, b))
                bufs.append(self.bufs[g][buf])
                self.steps_trained.append((ts, self.block_id(g, b)))
                if self.world > 1 or self.emulate_traffic:
                    free = (set(range(len(self.bufs[g]))) - set(self.where[g].values())).pop()
                    moves.append((g, b, buf, free, (b + 2) % self.nb))
            next_layouts = None
            if ts == self.nb - 1:   # the next boundary's layouts, in front of this epoch's last launch (see _plan_redeal)
                want_next = (epoch + 1) - (epoch + 1) % self.redeal_every
                if want_next != self.layout_epoch:
                    next_layouts = (want_next, self._fetch_layouts(self.layout_epoch, want_next))
            with self._timed("launch", self.stream):
                self.trainer.conveyor_enqueue(epoch, self.layout_epoch, blocks, bufs, lr, reg, use_bias, neg_population, flags)
            trained = None
            if self.stream is not None:
                trained = torch.cuda.Event()
                trained.record(self.stream)
        if self.world > 1:
            with self._on(self.comm):
                if self.comm is not None:
                    self.comm.wait_event(trained)
                else:                                 # gloo: the previous phase's sends still own the buffers these receives fill
                    for w in self._sent:
                        w.wait()
                ops = []
                for g, b, buf, free, nxt in moves:
                    s = self.strides[g]
                    dst, src = ranks[(self.rank - s) % self.world], ranks[(self.rank + s) % self.world]
                    ops.append(dist.P2POp(dist.isend, self.bufs[g][buf], dst, self.group))
                    ops.append(dist.P2POp(dist.irecv, self.bufs[g][free], src, self.group))
                with self._timed("move", self.comm):
                    works = [_OnceWork(w) for w in dist.batch_isend_irecv(ops)]
                    if self.comm is not None:
                        for w in works:
                            w.wait()                  # (stream-level on RCCL: the communication stream waits, the host does not)
                if self.comm is not None:
                    ev = torch.cuda.Event()
                    ev.record(self.comm)
                    for g, b, buf, free, nxt in moves:
                        self.arrived[g][free] = ev
                else:
                    for g, b, buf, free, nxt in moves:
                        self.arrived[g][free] = works # gloo: waited for before the block is trained (or gathered)
                    self._sent = list(works)
            for g, b, buf, free, nxt in moves:
                del self.where[g][b]
                self.where[g][nxt] = free
        elif self.emulate_traffic:
            # one rank: the block goes to the free buffer as if it travelled (and is trained from there next epoch)
            with self._on(self.comm):
                if self.comm is not None:
                    self.comm.wait_event(trained)
                with self._timed("move", self.comm):
                    for g, b, buf, free, nxt in moves:
                        self.bufs[g][free].copy_(self.bufs[g][buf])
                ev = None
                if self.comm is not None:
                    ev = torch.cuda.Event()
                    ev.record(self.comm)
                for g, b, buf, free, nxt in moves:
                    self.arrived[g][free] = ev
                    self.where[g][b] = free
        self.t += 1
        if next_layouts is not None:
            # the next boundary's plan, beside this epoch's last launch and transfers (the boundary then only moves rows)
            self._plan = self._plan_redeal(self.layout_epoch, next_layouts[0], layouts=next_layouts[1])

    def run_epoch(self, lr, reg, use_bias=True, neg_population=0, flags=0):
        for _ in range(self.nb):
            self.step(lr, reg, use_bias, neg_population, flags)

    def _drain(self):
        for g in range(self.K):
            for buf in range(len(self.bufs[g])):
                self._await(self.arrived[g][buf])
                self.arrived[g][buf] = None
        for w in self._sent:
            w.wait()
        self._sent = []
        if self.stream is not None:
            self.stream.synchronize()
            self.comm.synchronize()

    def finish(self):
        """(correct, skipped) of this rank since the last finish(); every transfer has landed"""
        self._drain()
        return self.trainer.sync()

    def gather(self):
        """the full (V, B) host tables on every rank; at an epoch boundary (every rank holds its home blocks)"""
        if self.t % self.nb:
            raise RuntimeError("gather() in the middle of an epoch (step %d of %d)" % (self.t % self.nb, self.nb))
        self._drain()
        home = self.home_blocks()
        mine = torch.stack([self.bufs[g][self.where[g][b]] for g, b, _ in home])
        if self.world > 1:
            comm_dev = _comm_device(self.device, self.group)
            full = torch.empty((self.world * len(home),) + tuple(mine.shape[1:]), dtype=mine.dtype, device=comm_dev)
            dist.all_gather_into_tensor(full, mine.to(comm_dev).contiguous(), group=self.group)
        else:
            full = mine
        full = full.cpu().numpy().reshape(self.world, len(home), -1)
        with self._on(self.stream):
            slot_item, _ = self._layout(self.layout_epoch)
            slot_item = slot_item.cpu().numpy()
        V = np.zeros((self.n_items, self.k), np.float32)
        B = np.zeros(self.n_items, np.float32)
        if self._tail is not None:
            V[self.n_train:], B[self.n_train:] = self._tail
        W, k = self.W, self.k
        for r in range(self.world):
            for h, (_, _, blk) in enumerate(self.home_blocks(r)):
                items = slot_item[blk * W: (blk + 1) * W]
                ok = items >= 0
                V[items[ok]] = full[r, h, : W * k].reshape(W, k)[ok]
                B[items[ok]] = full[r, h, W * k:][ok]
        return V, B

    def close(self):
        if self.trainer is not None:
            self.trainer.close()
        self.trainer = None


class _DeviceMfBlockTrainer:
    """the ratings of ONE item block of a rank (item ids local to the block) as a cornac_hip_mf handle that trains into the
    rank's shared user tables and into whichever buffer currently holds the block"""

    FORCE_FORM = None     # tools/bench_mf_rotation.py: 1 = the fused atomic kernel, 2 = the handle's own block rotation

    def __init__(self, rid, lid, val, n_users, rows, k, U, Bu, stream, device_index):
        from . import _lib

        self.t = _lib.MfTrainer(rid, lid, val, n_users, rows, k, device_index)
        # A block is a few hundred to a few thousand item rows and a step a few 10^4..10^7 ratings: in the plain fused atomic
        # kernel every row would take dozens to hundreds of CONCURRENT updates computed from one stale copy (most of a step's
        # ratings are in flight at once) and the factorisation diverges — measured: 6 000 users x 300 rows, 67 k ratings per
        # step, and 60 024 users x 1 111 rows, 0.78 M per step (a node rank's share of the Netflix shape): loss = inf.
        #   form 3: the fused kernel, its launch throttled to ~4 ratings in flight per item row and the rows that still take
        #           more than 32 concurrent updates trained through copies merged after the launch (cornac_hip.h) — the step
        #           handle's form: ~1 ms per step of 0.78 M ratings, 8 ms per step of 6.3 M
        #           (profiles/r06_mf_rotation_forms.log);
        #   form 2: the handle's own (user block x item bin) rotation, every update applied exactly once — 8 launches x 32
        #           barrier-separated sub-rounds whatever the size, and with ~1 000 rows every row is "hot" (no LDS bin,
        #           atomics): 4.4 ms and 11.8 ms for the same two steps.  It pays where the plain handle picks it: many rows
        #           (each a small share of the ratings) and >= 2^24 ratings per step (Netflix whole: 31 ms per 100 M).
        form = self.FORCE_FORM
        if form is None:
            form = 2 if (32 < k <= 256 and rows >= 8192 and len(val) >= (1 << 24)) else 3
        self.t.hogwild_form(form)
        self.t.bind_users(U.data_ptr(), Bu.data_ptr())
        if stream is not None:
            self.t.set_stream(stream.cuda_stream)
        self._keep = (U, Bu)

    def enqueue(self, V, Bi, lr, reg, mu, use_bias):
        self.t.bind_items(V.data_ptr(), Bi.data_ptr())
        self.t.epoch_enqueue(0, 1, lr, reg, mu, use_bias)

    def sync(self):
        return self.t.sync()

    def close(self):
        self.t.close()


class MfBlockRotationTrainer:
    """Multi-GPU MF (fit_sgd, backend_cpu.pyx:35-97), regime 2: block rotation.  A rating touches one user row and one item
    row (backend_cpu.pyx:62-88), so with the ratings partitioned BY USER over the ranks and the item table cut into 2 N
    blocks, rank r can train "its users x block b" while no other rank touches block b or r's users: in step t of an epoch
    rank r trains block (2 r + t) mod 2 N and hands it to rank r - 1, which needs it in step t + 2 — one whole step for the
    transfer, beside the next launch (the schedule of BinConveyorBprTrainer with one ring).  Every rating is applied exactly
    once per epoch to the one copy of its item row: nothing is reconciled, nothing is stale, and unlike BPR's conveyor there is
    no sampler to speak of — the ranks' parallel run equals the serial execution of the same steps (tests/test_dist_cpu.py).
    Inside a step a rank's own ratings of the block run through its handle in whatever form cornac_hip_mf_fit would pick
    (hogwild semantics, as on one GPU).

    Items go to blocks by their position in `item_order` (the popularity order all ranks share): position p -> block p mod 2N,
    row p div 2N — the popular items are dealt evenly.  One cornac_hip_mf handle per block of the rank (created over the
    rank's ratings of that block); they share the rank's U / Bu (cornac_hip_mf_bind_users) and are bound to the buffer that
    holds their block before every step (cornac_hip_mf_bind_items: no host wait on a re-bind).
    virtual_world (one rank only): lay the blocks out for that many ranks, all of them this rank's."""

    def __init__(self, rid, cid, val, n_users, n_items, k, device, group=None, trainer_factory=None, item_order=None,
                 emulate_traffic=False, virtual_world=None):
        self.device, self.group, self.k = device, group, int(k)
        self.world, self.rank = _world(group)
        self.lay_world = int(virtual_world) if (virtual_world and self.world == 1) else self.world
        self.nb = 2 * self.lay_world
        self.n_users, self.n_items = int(n_users), int(n_items)
        cuda = device.type == "cuda"
        self.stream = torch.cuda.Stream(device) if cuda else None
        self.comm = torch.cuda.Stream(device) if cuda else None
        self.emulate_traffic = bool(emulate_traffic) and self.world == 1
        order = np.arange(self.n_items, dtype=np.int64) if item_order is None else np.asarray(item_order, np.int64)
        pos = np.empty(self.n_items, np.int64)
        pos[order] = np.arange(self.n_items)
        self.item_block, self.item_row = pos % self.nb, pos // self.nb
        self.W = (self.n_items + self.nb - 1) // self.nb          # rows of a block buffer
        self.U = torch.zeros((self.n_users, self.k), dtype=torch.float32, device=device)
        self.Bu = torch.zeros(self.n_users, dtype=torch.float32, device=device)
        rid, cid, val = np.asarray(rid, np.int64), np.asarray(cid, np.int64), np.asarray(val, np.float32)
        self.nnz = len(val)
        blk = self.item_block[cid]
        self.trainers = []
        for b in range(self.nb):
            sel = np.flatnonzero(blk == b)                         # (stable: the stored order inside a block)
            if len(sel) == 0:
                self.trainers.append(None)
                continue
            args = (rid[sel], self.item_row[cid[sel]], val[sel], self.n_users, self.W, self.k, self.U, self.Bu)
            if trainer_factory is not None:
                self.trainers.append(trainer_factory(*args))
            else:
                self.trainers.append(_DeviceMfBlockTrainer(*args, self.stream, device.index or 0))
        n_home = self.nb if self.world == 1 else 2
        self.bufs = [torch.zeros(self.W * (self.k + 1), dtype=torch.float32, device=device) for _ in range(n_home + 1)]
        self.where = {b: b for b in range(self.nb)} if self.world == 1 else {2 * self.rank: 0, 2 * self.rank + 1: 1}
        self.arrived = [None] * (n_home + 1)
        self._sent = []
        self.t = 0
        self.steps_trained = []

    # ---- layout ----
    def home_blocks(self, rank=None):
        if self.world == 1:
            return list(range(self.nb))
        r = self.rank if rank is None else rank
        return [2 * r, 2 * r + 1]

    def _views(self, buf):
        flat = self.bufs[buf]
        return flat[: self.W * self.k].view(self.W, self.k), flat[self.W * self.k:]

    def _on(self, stream):
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def load_items(self, V, Bi):
        """the full host tables -> this rank's home blocks"""
        V, Bi = np.asarray(V, np.float32), np.asarray(Bi, np.float32)
        with self._on(self.stream):
            for b in self.home_blocks():
                items = np.flatnonzero(self.item_block == b)
                rows = torch.as_tensor(self.item_row[items], device=self.device)
                v, bb = self._views(self.where[b])
                v.zero_()
                bb.zero_()
                v.index_copy_(0, rows, torch.as_tensor(V[items]).to(self.device))
                bb.index_copy_(0, rows, torch.as_tensor(Bi[items]).to(self.device))
        if self.stream is not None:
            self.stream.synchronize()

    def set_user_factors(self, U, Bu):
        with self._on(self.stream):
            self.U.copy_(torch.as_tensor(np.asarray(U, np.float32)).to(self.device))
            self.Bu.copy_(torch.as_tensor(np.asarray(Bu, np.float32)).to(self.device))
        if self.stream is not None:
            self.stream.synchronize()

    def get_user_factors(self):
        self._drain()
        return self.U.cpu().numpy(), self.Bu.cpu().numpy()

    def _await(self, got):
        if got is None:
            return
        if isinstance(got, list):
            for w in got:
                w.wait()
        elif self.stream is not None:
            self.stream.wait_event(got)

    # ---- a step ----
    def step(self, lr, reg, mu, use_bias=True):
        ts = self.t % self.nb
        ranks = dist.get_process_group_ranks(self.group) if (self.group is not None and self.world > 1) else list(range(self.world))
        b = (2 * self.rank + ts) % self.nb
        buf = self.where[b]
        with self._on(self.stream):
            self._await(self.arrived[buf])
            self.arrived[buf] = None
            v, bb = self._views(buf)
            if self.trainers[b] is not None:
                self.trainers[b].enqueue(v, bb, lr, reg, mu, use_bias)
            self.steps_trained.append((ts, b))
            trained = None
            if self.stream is not None:
                trained = torch.cuda.Event()
                trained.record(self.stream)
        if self.world > 1:
            free = (set(range(len(self.bufs))) - set(self.where.values())).pop()
            nxt = (b + 2) % self.nb
            with self._on(self.comm):
                if self.comm is not None:
                    self.comm.wait_event(trained)
                else:
                    for w in self._sent:
                        w.wait()
                dst, src = ranks[(self.rank - 1) % self.world], ranks[(self.rank + 1) % self.world]
                ops = [dist.P2POp(dist.isend, self.bufs[buf], dst, self.group), dist.P2POp(dist.irecv, self.bufs[free], src, self.group)]
                works = [_OnceWork(w) for w in dist.batch_isend_irecv(ops)]
                if self.comm is not None:
                    for w in works:
                        w.wait()
                    ev = torch.cuda.Event()
                    ev.record(self.comm)
                    self.arrived[free] = ev
                else:
                    self.arrived[free] = works
                    self._sent = list(works)
            del self.where[b]
            self.where[nxt] = free
        elif self.emulate_traffic:
            free = (set(range(len(self.bufs))) - set(self.where.values())).pop()
            with self._on(self.comm):
                if self.comm is not None:
                    self.comm.wait_event(trained)
                self.bufs[free].copy_(self.bufs[buf])
                ev = None
                if self.comm is not None:
                    ev = torch.cuda.Event()
                    ev.record(self.comm)
                self.arrived[free] = ev
                self.where[b] = free
        self.t += 1

    def run_epoch(self, lr, reg, mu, use_bias=True):
        for _ in range(self.nb):
            self.step(lr, reg, mu, use_bias)

    def _drain(self):
        for buf in range(len(self.bufs)):
            self._await(self.arrived[buf])
            self.arrived[buf] = None
        for w in self._sent:
            w.wait()
        self._sent = []
        if self.stream is not None:
            self.stream.synchronize()
            self.comm.synchronize()

    def finish(self):
        """every transfer has landed; the sum of squared errors of this rank's ratings since the last finish() (0.5 x the
        all-reduced sum is the reference's epoch loss, backend_cpu.pyx:86-88)"""
        self._drain()
        return float(sum(t.sync() for t in self.trainers if t is not None))

    def gather(self):
        """the full (V, Bi) host tables on every rank; at an epoch boundary (every rank holds its home blocks)"""
        if self.t % self.nb:
            raise RuntimeError("gather() in the middle of an epoch (step %d of %d)" % (self.t % self.nb, self.nb))
        self._drain()
        home = self.home_blocks()
        mine = torch.stack([self.bufs[self.where[b]] for b in home])
        if self.world > 1:
            comm_dev = _comm_device(self.device, self.group)
            full = torch.empty((self.world * len(home),) + tuple(mine.shape[1:]), dtype=mine.dtype, device=comm_dev)
            dist.all_gather_into_tensor(full, mine.to(comm_dev).contiguous(), group=self.group)
        else:
            full = mine
        full = full.cpu().numpy().reshape(self.world, len(home), -1)
        V = np.zeros((self.n_items, self.k), np.float32)
        Bi = np.zeros(self.n_items, np.float32)
        W, k = self.W, self.k
        for r in range(self.world):
            for h, b in enumerate(self.home_blocks(r)):
                items = np.flatnonzero(self.item_block == b)
                rows = self.item_row[items]
                V[items] = full[r, h, : W * k].reshape(W, k)[rows]
                Bi[items] = full[r, h, W * k:][rows]
        return V, Bi

    def close(self):
        for t in self.trainers:
            if t is not None:
                t.close()
        self.trainers = []


# ---- model-level entry points: model.fit(train_set) over all ranks of a process group ----------------------------------
def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _comm_device(device, group):
    return device if (device is not None and device.type == "cuda") else torch.device("cpu")


def _broadcast_from_rank0(arrays, device, group):
    """in place: every rank's arrays become rank 0's (identical initial tables whatever the ranks' generators did)"""
    if _world(group)[0] == 1:
        return
    for a in arrays:
        t = torch.as_tensor(np.ascontiguousarray(a)).to(_comm_device(device, group))
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        a[...] = t.cpu().numpy()


def _gather_user_rows(local, bounds, device, group):
    """rows [bounds[r], bounds[r+1]) live on rank r: every rank receives the full array (one padded all_gather)"""
    world, _ = _world(group)
    local = np.ascontiguousarray(local)
    if world == 1:
        return local
    width = local.shape[1:] if local.ndim > 1 else ()
    cap = int(np.max(np.diff(bounds)))
    pad = torch.zeros((cap,) + tuple(width), dtype=torch.as_tensor(local).dtype, device=_comm_device(device, group))
    pad[: len(local)] = torch.as_tensor(local)
    out = torch.empty((world * cap,) + tuple(width), dtype=pad.dtype, device=pad.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.cpu().numpy().reshape((world, cap) + tuple(width))
    return np.concatenate([out[r, : int(bounds[r + 1] - bounds[r])] for r in range(world)])


def _sum_over_ranks(values, device, group):
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=_comm_device(device, group))
    if _world(group)[0] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [float(x) for x in t.cpu()]


def global_item_degrees(indices_local, n_items, device=None, group=None):
    """item degrees of the WHOLE matrix: the ranks all-reduce the degrees among their own users (n_items counts — 107 KB at
    the ML-20M shape)"""
    deg = torch.as_tensor(np.bincount(np.asarray(indices_local, np.int64), minlength=int(n_items)).astype(np.int64))
    if _world(group)[0] > 1:
        deg = deg.to(_comm_device(device, group))
        dist.all_reduce(deg, op=dist.ReduceOp.SUM, group=group)
    return deg.cpu().numpy()


def population_from_degrees(deg, at_most=1 << 26):
    """the multiset a popularity-weighted draw picks from: id i repeated deg[i] times, scaled down to at most `at_most`
    entries (every id with a positive degree at least once)"""
    deg = np.asarray(deg, np.int64)
    total = int(deg.sum())
    if total > at_most:
        deg = np.where(deg > 0, np.maximum(1, np.rint(deg * (float(at_most) / total)).astype(np.int64)), 0)
    return np.repeat(np.arange(len(deg), dtype=np.int32), deg)


def global_negative_population(indices_local, n_items, device=None, group=None, at_most=1 << 26):
    """WBPR's negative population over all ranks (recom_wbpr.pyx:135: neg_item_ids = X.indices of the WHOLE matrix): every
    rank builds the same multiset from the all-reduced item degrees, item i repeated degree(i) times (scaled to at most
    `at_most` entries, every interacted item at least once)"""
    return population_from_degrees(global_item_degrees(indices_local, n_items, device, group), at_most)


def fit_bpr_sharded(model, train_set, device=None, group=None, sync_per_epoch=None, sparse_threshold=None,
                    trainer_factory=None, local_popularity=False, rule=None, regime="auto", rings=1):
    """`model.fit(train_set)` for a cornac_amd BPR / WBPR over all ranks of the process group (regime 1).  Every rank
    calls it with a model built from the same arguments and the SAME train_set; the users are cut into contiguous ranges
    of equal interaction counts, rank r trains its range in hogwild mode against its replica of the item table
    (ShardedBprTrainer), and on return every rank's model holds the complete u_factors / i_factors / i_biases.
    The reference has no counterpart (single process); seeded SEQUENTIAL semantics do not shard, so the model must be in
    hogwild mode (`mode="hogwild"`, or no seed) — a seed then fixes the initial tables and the sample streams.
    sync_per_epoch = None: exchange_schedule() of the largest rank's interaction count (the same on every rank): how many
    exchanges per epoch — or, for sparse item sides, every how many epochs — and the reconciliation rule (rule = None).
    WBPR draws its negative as the item of a uniformly chosen interaction (recom_wbpr.pyx:135 reads X.indices: popularity-
    weighted).  Over several ranks the population is the GLOBAL one: the ranks all-reduce their item degrees (n_items
    counts) and every handle is given a population with those multiplicities (global_negative_population,
    cornac_hip_bpr_set_negative_population; the draw then runs in the fused kernel).  local_popularity=True keeps each
    rank's own interactions as its population (the LDS-bin form's binned draw; the popularity of the rank's users only).
    trainer_factory(table, indptr, indices, n_local, n_items, total_items, k): test hook (host stand-ins on gloo).
    regime: "replicated" = this regime; "ring" = fit_bpr_ring (its sampling contract: that docstring); "auto" (default) = the
    ring for sparse item sides (prefers_conveyor — where the replicas lag in mid-training), the replicas
    otherwise and whenever the caller fixes the replica protocol (sync_per_epoch, rule,
    sparse_threshold, trainer_factory).  rings: strided rings of the conveyor (fit_bpr_ring)."""
    from . import _lib
    from .recommender import Recommender

    if regime not in ("auto", "replicated", "ring"):
        raise ValueError("regime must be 'auto', 'replicated' or 'ring'")
    if model.effective_mode != "hogwild":
        raise ValueError("sequential (seeded, mode=None) semantics do not shard: build the model with mode='hogwild'")
    world, rank = _world(group)
    device = device if device is not None else torch.device("cpu")
    if regime == "auto" and (sync_per_epoch is not None or rule is not None or sparse_threshold is not None
                             or trainer_factory is not None):
        regime = "replicated"
    if regime == "auto" and world > 1 and model._neg_population == _lib.NEG_POPULARITY and not local_popularity:
        regime = "replicated"   # WBPR with the GLOBAL popularity: the conveyor only has the rank-local one (fit_bpr_ring)
    if regime == "ring" and trainer_factory is not None:
        raise ValueError("trainer_factory is the replicated regime's test hook (its signature differs from the conveyor's): "
                         "call fit_bpr_ring(..., trainer_factory=...) for the ring")
    if regime != "replicated":
        X0 = train_set.matrix
        per_rank = int(X0.nnz // max(world, 1)) + 1
        if regime == "ring" or prefers_conveyor(per_rank, train_set.num_items):
            return fit_bpr_ring(model, train_set, device=device, group=group,
                                local_popularity=local_popularity or model._neg_population != _lib.NEG_POPULARITY, rings=rings)
    Recommender.fit(model, train_set)
    model._init()
    if model.trains_float64:
        raise ValueError("float64 tables train on the sequential engine only")
    _broadcast_from_rank0([model.u_factors, model.i_factors, model.i_biases], device, group)
    X = train_set.matrix
    if not X.has_sorted_indices:
        X.sort_indices()
    bounds = partition_users_by_nnz(X.indptr, world)
    counts = [int(X.indptr[bounds[r + 1]] - X.indptr[bounds[r]]) for r in range(world)]
    if min(counts) == 0:
        raise ValueError("a rank would receive no interactions (%r): use fewer ranks" % (counts,))
    u0, u1 = int(bounds[rank]), int(bounds[rank + 1])
    indptr, indices = slice_csr(X.indptr, X.indices, u0, u1)
    n_local, nnz = u1 - u0, counts[rank]
    population = None
    if world > 1 and model._neg_population == _lib.NEG_POPULARITY and not local_popularity:
        population = global_negative_population(indices, train_set.num_items, device, group)
    interval = 1
    if sync_per_epoch is None:
        sync_per_epoch, interval, scheduled_rule = exchange_schedule(max(counts), model.total_items)
        rule = rule if rule is not None else scheduled_rule
    if rule is None:
        rule = "sqrt" if int(sync_per_epoch) >= 16 else "align"
    if sparse_threshold is not None:
        interval = 1
    parts = max(1, min(int(sync_per_epoch), min(counts)))
    if trainer_factory is None:
        trainer = _lib.BprTrainer(indptr, indices, n_local, train_set.num_items, n_local, model.total_items, model.k,
                                  device=device.index or 0)
        sh = ShardedBprTrainer(trainer, model.total_items, model.k, device, sync_every=nnz, group=group,
                               sparse_threshold=sparse_threshold, rule=rule)
    else:
        sh = ShardedBprTrainer(None, model.total_items, model.k, device, sync_every=nnz, group=group,
                               sparse_threshold=sparse_threshold, rule=rule)
        trainer = sh.trainer = trainer_factory(sh.table, indptr, indices, n_local, train_set.num_items, model.total_items,
                                               model.k)
    try:
        trainer.set_factors(model.u_factors[u0:u1], None, None)
        if population is not None:
            trainer.set_negative_population(population)
        lo, hi = int(model.rng.randint(2 ** 31)), int(model.rng.randint(2 ** 31))
        trainer.seed_hogwild((((hi << 32) | lo) + 7919 * rank) & 0xFFFFFFFFFFFFFFFF)
        sh.load_items(model.i_factors, model.i_biases)
        # every rank must take the same protocol (the same number of collectives in the same order): the resident
        # exchange only when every rank's trainer has it for its slice
        resident = bool(_sum_over_ranks([0 if 1 <= parts <= 32 and sh.resident_bins(model._neg_population) > 0 else 1],
                                        device, group)[0] == 0)
        for _ in range(model.max_iter):
            sh.run_epoch(nnz, parts, model.learning_rate, model.lambda_reg, model.use_bias, model._neg_population,
                         resident=resident, epochs_per_exchange=interval)
        correct, skipped = sh.finish()
        U_local = trainer.get_user_factors()
        V, B = sh.table.V.cpu().numpy(), sh.table.B.cpu().numpy()
    finally:
        trainer.close()
    model.u_factors[: bounds[-1]] = _gather_user_rows(U_local, bounds, device, group)
    model.i_factors[...] = V
    model.i_biases[...] = B
    c, s = _sum_over_ranks([correct, skipped], device, group)
    model.fit_stats = [(int(c), int(s))]
    model._drop_scorer()
    return model


def fit_bpr_ring(model, train_set, device=None, group=None, trainer_factory=None, local_popularity=True, rings=1,
                 redeal_every=1):
    """`model.fit(train_set)` for a cornac_amd BPR / WBPR over all ranks of the process group with the item table sharded
    by row and rotating around the ring (regime 2, BinConveyorBprTrainer): the same calling convention and restrictions as
    fit_bpr_sharded, no replica and no reconciliation rule — every item row is in one place at any time, so the result
    is the serial execution of the ranks' steps.  The memory an item table needs per rank is 3 / (2 N) of it.

    Sampling contract: every interaction is drawn with the reference's probability 1 / nnz per draw, nnz draws per epoch
    (each by the rank that owns its user); the negative of a draw is uniform over the ~cap items that share the positive's
    LDS bin in that epoch; bins — and with them the conveyor's blocks, which are ranges of bins — are re-dealt from ALL train
    items every `redeal_every` epochs with a key the ranks share, over the popularity order of the WHOLE matrix (the ranks
    all-reduce their item degrees), so every (positive, negative) pair of items can meet (recom_bpr.pyx:235-238 draws j over
    all items; csrc/bpr_ldsbin.inc, tests/test_ldsbin_deal_cpu.py).
    WBPR: the popularity-weighted negative is the item of a second interaction drawn among the RANK'S OWN users' interactions
    with the bin's items (local popularity — the only form the conveyor has: local_popularity=False raises)."""
    from . import _lib
    from .recommender import Recommender

    if model.effective_mode != "hogwild":
        raise ValueError("sequential (seeded, mode=None) semantics do not shard: build the model with mode='hogwild'")
    world, rank = _world(group)
    device = device if device is not None else torch.device("cpu")
    Recommender.fit(model, train_set)
    model._init()
    if model.trains_float64:
        raise ValueError("float64 tables train on the sequential engine only")
    if world > 1 and model._neg_population == _lib.NEG_POPULARITY and not local_popularity:
        raise ValueError("the conveyor draws WBPR's negative by the popularity among the rank's own users (local_popularity=True); "
                         "the global popularity is served by regime='replicated'")
    seeds = np.array([int(model.rng.randint(2 ** 31)), int(model.rng.randint(2 ** 31))], np.int64)
    _broadcast_from_rank0([model.u_factors, model.i_factors, model.i_biases, seeds], device, group)
    X = train_set.matrix
    if not X.has_sorted_indices:
        X.sort_indices()
    bounds = partition_users_by_nnz(X.indptr, world)
    counts = [int(X.indptr[bounds[r + 1]] - X.indptr[bounds[r]]) for r in range(world)]
    if min(counts) == 0:
        raise ValueError("a rank would receive no interactions (%r): use fewer ranks" % (counts,))
    u0, u1 = int(bounds[rank]), int(bounds[rank + 1])
    indptr, indices = slice_csr(X.indptr, X.indices, u0, u1)
    # the popularity order of the WHOLE matrix (every rank must deal the same items to the same bins)
    deg = global_item_degrees(indices, train_set.num_items, device, group)
    order = np.argsort(-deg, kind="stable").astype(np.int32)
    ring = BinConveyorBprTrainer(indptr, indices, u1 - u0, model.total_items, model.k, device, group=group,
                                 trainer_factory=trainer_factory, seed=(int(seeds[1]) << 32) | int(seeds[0]), rings=rings,
                                 n_train_items=train_set.num_items, item_order=order, redeal_every=redeal_every)
    try:
        ring.set_user_factors(model.u_factors[u0:u1])
        ring.load_items(model.i_factors, model.i_biases)
        for _ in range(model.max_iter):
            ring.run_epoch(model.learning_rate, model.lambda_reg, model.use_bias, model._neg_population)
        correct, skipped = ring.finish()
        U_local = ring.get_user_factors()
        V, B = ring.gather()
    finally:
        ring.close()
    model.u_factors[: bounds[-1]] = _gather_user_rows(U_local, bounds, device, group)
    model.i_factors[...] = V
    model.i_biases[...] = B
    c, s = _sum_over_ranks([correct, skipped], device, group)
    model.fit_stats = [(int(c), int(s))]
    model._drop_scorer()
    return model


def fit_mf_sharded(model, train_set, device=None, group=None, parts_per_epoch=None, sparse_threshold=None,
                   trainer_factory=None, rule="align", regime="auto", block_trainer_factory=None):
    """`model.fit(train_set)` for a cornac_amd MF (backend "hip", hogwild mode) over all ranks of the process group:
    users — with their ratings, in stored order — cut into contiguous ranges of equal rating counts; every rank returns
    with the complete model.  Same calling convention and restrictions as fit_bpr_sharded; `early_stop` is not supported
    (it would need the global loss every epoch on the host).  model.loss_history holds 0.5 x the summed squared error of all
    ranks per epoch.
    regime: "rotation" — the item table in 2 N blocks that rotate over the ranks, every rating applied exactly once per epoch
    to the one copy of its item row (MfBlockRotationTrainer; equal to one process within 0.5 % of held-out RMSE in
    mid-training, tests/test_sharded_gpu.py); "replicated" — the item side replicated and reconciled (ShardedMfTrainer;
    measured on the device at R = 8 and the Netflix density: held-out RMSE 0.857 where one process has 0.531 after 4 epochs —
    the shared rows learn at a fraction of the pace); "auto" (default): rotation from 16 item rows per block on, else
    replicated.
    trainer_factory(table, rid_local, cid, val, n_local, n_items, k): test hook of the replicated regime (implies it);
    block_trainer_factory: MfBlockRotationTrainer's."""
    from . import _lib
    from .recommender import Recommender

    if model.effective_mode != "hogwild":
        raise ValueError("sequential (seeded, mode=None) semantics do not shard: build the model with mode='hogwild'")
    if getattr(model, "early_stop", False):
        raise ValueError("early_stop is not supported by the sharded fit")
    if trainer_factory is None and getattr(model, "backend", "hip") != "hip":
        raise ValueError("fit_mf_sharded drives the 'hip' backend (fit_sgd); this model was built with backend=%r" % (model.backend,))
    if not getattr(model, "trainable", True):
        raise ValueError("the model is not trainable (trainable=False): nothing to fit")
    world, rank = _world(group)
    device = device if device is not None else torch.device("cpu")
    Recommender.fit(model, train_set)
    model._init()
    _broadcast_from_rank0([model.u_factors, model.i_factors, model.u_biases, model.i_biases], device, group)
    X = train_set.matrix
    bounds = partition_users_by_nnz(X.indptr, world)
    counts = [int(X.indptr[bounds[r + 1]] - X.indptr[bounds[r]]) for r in range(world)]
    if min(counts) == 0:
        raise ValueError("a rank would receive no ratings (%r): use fewer ranks" % (counts,))
    u0, u1 = int(bounds[rank]), int(bounds[rank + 1])
    rid, cid, val = train_set.uir_tuple
    mine = (rid >= u0) & (rid < u1)
    rid_l, cid_l, val_l = (rid[mine] - u0).astype(np.int64), cid[mine].astype(np.int64), val[mine].astype(np.float32)
    n_local = u1 - u0
    mu = float(model.global_mean)
    if regime not in ("auto", "rotation", "replicated"):
        raise ValueError("regime must be 'auto', 'rotation' or 'replicated', not %r" % (regime,))
    if regime == "auto":
        ok = model.num_items // (2 * world) >= 16
        regime = "rotation" if (block_trainer_factory is not None or (trainer_factory is None and ok)) else "replicated"
    if regime == "rotation":
        order = np.argsort(-np.asarray(global_item_degrees(cid_l, model.num_items, device, group)), kind="stable")
        rot = MfBlockRotationTrainer(rid_l, cid_l, val_l, n_local, model.num_items, model.k, device, group=group,
                                     trainer_factory=block_trainer_factory, item_order=order)
        try:
            rot.set_user_factors(model.u_factors[u0:u1], model.u_biases[u0:u1])
            rot.load_items(model.i_factors[: model.num_items], model.i_biases[: model.num_items])
            losses = []
            for _ in range(model.max_iter):
                rot.run_epoch(model.learning_rate, model.lambda_reg, mu, model.use_bias)
                losses.append(0.5 * _sum_over_ranks([rot.finish()], device, group)[0])
                if not np.isfinite(losses[-1]):
                    raise FloatingPointError("the sharded MF fit diverged: non-finite loss in epoch %d" % len(losses))
            U_local, Bu_local = rot.get_user_factors()
            V, Bi = rot.gather()
        finally:
            rot.close()
        model.u_factors[: bounds[-1]] = _gather_user_rows(U_local, bounds, device, group)
        model.u_biases[: bounds[-1]] = _gather_user_rows(Bu_local, bounds, device, group)
        model.i_factors[: model.num_items] = V
        model.i_biases[: model.num_items] = Bi
        model.loss_history, model.epochs_run = np.asarray(losses, np.float32), len(losses)
        model._drop_scorer()
        return model
    if trainer_factory is None:
        trainer = _lib.MfTrainer(rid_l, cid_l, val_l, n_local, model.num_items, model.k, device=device.index or 0)
        sh = ShardedMfTrainer(trainer, model.num_items, model.k, device, parts_per_epoch=parts_per_epoch, group=group,
                              sparse_threshold=sparse_threshold, rule=rule)
    else:
        sh = ShardedMfTrainer(None, model.num_items, model.k, device, parts_per_epoch=parts_per_epoch, group=group,
                              sparse_threshold=sparse_threshold, rule=rule)
        trainer = sh.trainer = trainer_factory(sh.table, rid_l, cid_l, val_l, n_local, model.num_items, model.k)
    try:
        trainer.set_factors(model.u_factors[u0:u1], None, model.u_biases[u0:u1], None)
        sh.load_items(model.i_factors[: model.num_items], model.i_biases[: model.num_items])
        losses = []
        for _ in range(model.max_iter):
            sh.run_epoch(model.learning_rate, model.lambda_reg, mu, model.use_bias)
            losses.append(0.5 * _sum_over_ranks([sh.finish()], device, group)[0])
            if not np.isfinite(losses[-1]):  # (the sum is the same on every rank: all of them raise)
                raise FloatingPointError("the sharded MF fit diverged: non-finite loss in epoch %d (learning rate %g too "
                                         "large for the hottest item rows?)" % (len(losses), model.learning_rate))
        U_local, _, Bu_local, _ = trainer.get_factors()
        V, Bi = sh.table.V.cpu().numpy(), sh.table.B.cpu().numpy()
    finally:
        trainer.close()
    model.u_factors[: bounds[-1]] = _gather_user_rows(U_local, bounds, device, group)
    model.u_biases[: bounds[-1]] = _gather_user_rows(Bu_local, bounds, device, group)
    model.i_factors[: model.num_items] = V
    model.i_biases[: model.num_items] = Bi
    model.loss_history, model.epochs_run = np.asarray(losses, np.float32), len(losses)
    model._drop_scorer()
    return model


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
    dev = device if device is not None else torch.device("cpu")
    # the ranks agree on the list width before padding: rank_fn may clamp k (to the item count) and a rank with an empty
    # block has nothing to read the width from — MAX over the ranks' widths, 0 from an empty one
    w = torch.tensor([items.shape[1] if len(mine) else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
    width = int(w.item()) or k
    if len(mine) and items.shape[1] != width:
        grow = width - items.shape[1]
        items = np.concatenate([items, np.full((len(mine), grow), -1, np.int32)], axis=1)
        scores = np.concatenate([scores, np.full((len(mine), grow), -np.inf, np.float32)], axis=1)
    cap = int((cuts[1:] - cuts[:-1]).max())
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
