#!/usr/bin/env python3
"""Diagnostic of the resident exchange on one GPU (no process group): trains E epochs with n_ex exchange points and checks
the bookkeeping the protocol promises — bucket == keep (one rank, no twin), the table moved by exactly the sum of the
published deltas (+ what is still unpublished on hot rows), no correction differs from zero — and compares the result
with the plain launch of the same seed, per table."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cornac_amd import _lib, synth
from cornac_amd.dist import ShardedBprTrainer

k, epochs, n_ex = 64, int(os.environ.get("EPOCHS", "3")), int(os.environ.get("NEX", "16"))
nu, ni = 4000, 12800
users, items = synth.zipf_interactions(nu, ni, 600_000, 0.6, 4)
indptr, indices = synth.csr_from_sorted(users, items, nu)
nnz = len(indices)
rs = np.random.RandomState(4)
U0, V0, B0 = rs.normal(0, .1, (nu, k)).astype(np.float32), rs.normal(0, .1, (ni, k)).astype(np.float32), rs.normal(0, .1, ni).astype(np.float32)
dev = torch.device("cuda", 0)


def plain():
    tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
    tr.set_factors(U0, V0, B0); tr.seed_hogwild(5)
    cs = tr.fit_epochs(epochs, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    out = tr.get_factors(); tr.close()
    return cs, out


def cos(a, b, i):
    x, y = (a - i).ravel().astype(np.float64), (b - i).ravel().astype(np.float64)
    return float(x @ y / np.linalg.norm(x) / np.linalg.norm(y))


(cp, sp), (Up, Vp, Bp) = plain()
(_, _), (Uq, Vq, Bq) = plain()
print("plain vs plain: cos V %.5f B %.5f U %.5f" % (cos(Vq, Vp, V0), cos(Bq, Bp, B0), cos(Uq, Up, U0)))
tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
tr.set_factors(U0, None, None); tr.seed_hogwild(5)
sh = ShardedBprTrainer(tr, ni, k, dev, sync_every=nnz)
sh.load_items(V0, B0)
width = ni * k + ni
prev = np.concatenate([V0.ravel(), B0])
for e in range(epochs):
    sh.run_epoch(nnz, n_ex, 0.05, 0.01, resident=True)
    sh.stream.synchronize(); sh._resident["comm"].synchronize()
    flat, base = sh.table.flat.cpu().numpy(), sh.table.base.cpu().numpy()
    buckets, keeps = sh._resident["buckets"].cpu().numpy(), sh._resident["keeps"].cpu().numpy()
    applied = sh._resident["applied"].cpu().numpy()
    moved = base - prev                                   # published progress of this epoch
    pub = keeps.astype(np.float64).sum(0)
    print("epoch %d: bucket==keep %s | max |moved - sum keeps| %.3g (max |moved| %.3g) | rows with flat != base %d | applied in launch: min %d mean %.2f max %d"
          % (e, np.array_equal(buckets[:, :width], keeps), np.abs(moved - pub).max(), np.abs(moved).max(),
             len(np.unique(np.nonzero((flat != base)[: ni * k])[0] // k)), applied.min(), applied.mean(), applied.max()))
    prev = base.copy()
c, s = sh.finish()
V, B, U = sh.table.V.cpu().numpy(), sh.table.B.cpu().numpy(), tr.get_user_factors()
print("resident vs plain: cos V %.5f B %.5f U %.5f | correct %d vs %d skipped %d vs %d | lock timeouts %d"
      % (cos(V, Vp, V0), cos(B, Bp, B0), cos(U, Up, U0), c, cp, s, sp, tr.ldsbin_stats()["lock_timeouts"]))
tr.close()

# the established chunk protocol (16 launches per epoch with the overlapped exchange between them): another schedule of the
# same draws — how far does a different interleaving alone move the result?
tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
tr.set_factors(U0, None, None); tr.seed_hogwild(5)
sh = ShardedBprTrainer(tr, ni, k, dev, sync_every=nnz)
sh.load_items(V0, B0)
for e in range(epochs):
    sh.run_epoch(nnz, n_ex, 0.05, 0.01, resident=False)
c2, s2 = sh.finish()
V2, B2, U2 = sh.table.V.cpu().numpy(), sh.table.B.cpu().numpy(), tr.get_user_factors()
print("chunks vs plain:   cos V %.5f B %.5f U %.5f | correct %d skipped %d" % (cos(V2, Vp, V0), cos(B2, Bp, B0), cos(U2, Up, U0), c2, s2))
print("chunks vs resident: cos V %.5f B %.5f U %.5f" % (cos(V2, V, V0), cos(B2, B, B0), cos(U2, U, U0)))
tr.close()
