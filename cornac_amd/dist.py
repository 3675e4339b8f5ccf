"""Multi-GPU BPR: one process per GPU, users partitioned across ranks, item table replicated and
reconciled with RCCL all-reduce of its deltas (regime 1 "item table fits per GPU").

  * every rank owns a disjoint user population (its CSR slice and its U rows) -> user rows never
    leave the GPU and never conflict across GPUs: no data-path collective for them;
  * V and B are replicated; every `sync_every` samples each rank all-reduces (sum) its local delta
    (V - V_base, B - B_base) as ONE flat fp32 bucket and rebases — mathematically the ranks apply
    each other's SGD steps with a bounded delay, the same asynchrony class as Hogwild;
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
    """Flat [V | B] buffer + base copy + all-reduce-of-deltas rebase."""

    def __init__(self, total_items, k, device, group=None):
        self.total_items, self.k = int(total_items), int(k)
        n = self.total_items * self.k + self.total_items
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.base = torch.zeros(n, dtype=torch.float32, device=device)
        self.group = group

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
        """flat <- base + sum_over_ranks(flat - base); base <- flat"""
        if not (dist.is_available() and dist.is_initialized()):
            self.base.copy_(self.flat)
            return
        delta = self.flat - self.base
        dist.all_reduce(delta, op=dist.ReduceOp.SUM, group=self.group)
        self.base.add_(delta)
        self.flat.copy_(self.base)


class ShardedBprTrainer:
    """Drives one rank's cornac_hip BPR handle plus the replicated item table."""

    def __init__(self, trainer, total_items, k, device, sync_every, group=None):
        self.trainer = trainer
        self.table = ItemTableReplica(total_items, k, device, group)
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
        """enqueue n_samples hogwild samples in sync_every-sized chunks with a table sync after each"""
        left = int(n_samples)
        with self._on_stream():
            while left > 0:
                n = min(left, self.sync_every)
                self.trainer.hogwild_enqueue(n, lr, reg, use_bias, neg_population, flags)
                self.table.sync()
                left -= n

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
