#!/usr/bin/env python3
"""How often must the replicated item table be exchanged when a rank's interactions are SPARSE over the items — the
configs[4] slice: 62.5 M interactions per rank over 10 M item rows = 12.5 item-row updates per row and EPOCH?  The rule of
cornac_amd.dist (exchanges_per_epoch) keeps the staleness the ML-20M emulations found sufficient — 93.5 updates per row
and exchange — which at this density is one exchange every 7.5 EPOCHS.  This tool checks that on the CPU (no GPU
needed): R virtual ranks, each its own users, one Zipf item popularity, the density of the configs[4] slice scaled down;
every rank trains its replica with the oracle's BPR arithmetic (oracle_bpr_epoch_seq), and every `interval` epochs the
replicas are reconciled with exactly the algebra of ItemTableReplica (sqrt rule, the sum landing one interval late as
in the overlapped schedule).  Prints, per configuration, the pairwise accuracy of the CONSOLIDATED item table on probe
triplets of rank 0's users next to one process training on rank 0's data alone and one process on ALL data.

TEST INFRASTRUCTURE (it drives the oracle): nothing under cornac_amd/ imports it."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--items", type=int, default=40000)
ap.add_argument("--users", type=int, default=25000, help="per rank")
ap.add_argument("--per-user", type=int, default=10)
ap.add_argument("--k", type=int, default=16)
ap.add_argument("--epochs", type=int, default=24)
ap.add_argument("--intervals", default="1,2,4,8", help="epochs per exchange")
ap.add_argument("--zipf", type=float, default=0.8)
ap.add_argument("--rules", default="sqrt,align", help="reconciliation rules to run (ItemTableReplica: sqrt, align)")
args = ap.parse_args()
R, ni, nu, k, lr, reg = args.ranks, args.items, args.users, args.k, 0.05, 0.01


def make_rank(r):
    rs = np.random.RandomState(100 + r)
    p = 1.0 / np.arange(1, ni + 1) ** args.zipf
    p /= p.sum()
    # users prefer a window of the catalogue (their "taste") on top of the global popularity: there is something to learn
    taste = rs.randint(0, 16, nu)
    rows = []
    grp = (np.arange(ni) * 2654435761 % 16).astype(np.int64)
    for u in range(nu):
        w = p * np.where(grp == taste[u], 8.0, 1.0)
        rows.append(np.sort(rs.choice(ni, args.per_user, replace=False, p=w / w.sum())))
    indptr = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int32)
    indices = np.concatenate(rows).astype(np.int32)
    return indptr, indices


class Rank:
    def __init__(self, r, indptr, indices, V0):
        self.indptr, self.indices = indptr, indices
        self.user_ids = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr)).astype(np.int32)
        self.neg_ids = np.arange(ni, dtype=np.int32)
        self.U0 = ((np.random.RandomState(500 + r).uniform(0, 1, (len(indptr) - 1, k)).astype(np.float32) - 0.5) / k)
        self.r = r
        self.reset(V0)

    def reset(self, V0):
        self.U = self.U0.copy()
        self.V, self.B = V0.copy(), np.zeros(ni, np.float32)
        self.baseV, self.baseB = self.V.copy(), self.B.copy()
        self.local = None
        self.gp, self.gn = orc.MT19937(1000 + self.r), orc.MT19937(2000 + self.r)

    def epoch(self):
        c, s = C.c_int64(), C.c_int64()
        nnz = len(self.user_ids)
        rc = orc.lib().oracle_bpr_epoch_seq(self.gp.ptr, self.gn.ptr, nnz - 1, ni - 1, nnz, self.user_ids, self.indices,
                                            self.neg_ids, self.indptr, self.U, self.V, self.B, k, lr, reg, 1, C.byref(c),
                                            C.byref(s), None, None, None)
        assert rc == 0


def accuracy(U, V, B, probe):
    u, i, j = probe
    s = (U[u] * (V[i] - V[j])).sum(1) + B[i] - B[j]
    return float((s > 0).mean())


data = [make_rank(r) for r in range(R)]
nnz = len(data[0][1])
print("ranks %d, %d items, %d users x %d per rank: %.1f item-row updates per row and epoch (configs[4] slice: 12.5); "
      "the 93.5-updates rule asks for one exchange every %.1f epochs"
      % (R, ni, nu, args.per_user, 2.0 * nnz / ni, 93.5 * ni / (2.0 * nnz)), flush=True)
V0 = ((np.random.RandomState(7).uniform(0, 1, (ni, k)).astype(np.float32) - 0.5) / k)
ranks = [Rank(r, data[r][0], data[r][1], V0) for r in range(R)]
prs = np.random.RandomState(99)
ip0, ix0 = data[0]
pp = prs.randint(0, len(ix0), 100_000)
pu, pi = np.repeat(np.arange(nu), np.diff(ip0))[pp], ix0[pp]
pj = prs.randint(0, ni, len(pp))
ok = np.array([pj[t] not in ix0[ip0[pu[t]]:ip0[pu[t] + 1]] for t in range(len(pp))])
probe = (pu[ok], pi[ok], pj[ok])


def reconcile(ranks, pending, rule):
    """finish the pending exchange (its sum lands now), begin the next one: ItemTableReplica.step_sync for all ranks"""
    if pending is not None:
        SV, SB, cV, cB = pending
        if rule == "sqrt":
            RV = SV / np.sqrt(np.maximum(cV, 1.0))[:, None]
            RB = SB / np.sqrt(np.maximum(cB, 1.0))
        else:  # align: S * min(1, sum |d_r|^2 / |S|^2)
            n2 = (SV * SV).sum(1)
            RV = SV * np.where(n2 > 0, np.minimum(1.0, cV / np.maximum(n2, 1e-38)), 1.0)[:, None].astype(np.float32)
            RB = SB * np.where(SB * SB > 0, np.minimum(1.0, cB / np.maximum(SB * SB, 1e-38)), 1.0).astype(np.float32)
        for x in ranks:
            pV, pB = (x.V - x.baseV) - x.local[0], (x.B - x.baseB) - x.local[1]
            x.baseV += RV
            x.baseB += RB
            x.V[...] = x.baseV + pV
            x.B[...] = x.baseB + pB
    SV, SB = np.zeros((ni, k), np.float32), np.zeros(ni, np.float32)
    cV, cB = np.zeros(ni, np.float32), np.zeros(ni, np.float32)
    for x in ranks:
        dV, dB = x.V - x.baseV, x.B - x.baseB
        x.local = (dV, dB)
        SV += dV
        SB += dB
        if rule == "sqrt":
            cV += (dV != 0).any(1)
            cB += dB != 0
        else:
            cV += (dV * dV).sum(1)
            cB += dB * dB
    return SV, SB, cV, cB


# references: rank 0 alone (no exchange at all), and one process on every rank's data in turn (the item table sees R x
# the updates per epoch: what a perfect exchange would give)
ranks[0].reset(V0)
alone = []
for e in range(args.epochs):
    ranks[0].epoch()
    alone.append(accuracy(ranks[0].U, ranks[0].V, ranks[0].B, probe))
print("rank 0 alone:              " + " ".join("%.3f" % a for a in alone[3::4]), flush=True)
for x in ranks:
    x.reset(V0)
shared_V, shared_B = V0.copy(), np.zeros(ni, np.float32)
allacc = []
for e in range(args.epochs):
    for x in ranks:
        x.V, x.B = shared_V, shared_B
        x.epoch()
    allacc.append(accuracy(ranks[0].U, shared_V, shared_B, probe))
print("one process, all ranks' data: " + " ".join("%.3f" % a for a in allacc[3::4]), flush=True)

for rule, interval in [(r_, int(x)) for r_ in args.rules.split(",") for x in args.intervals.split(",")]:
    for x in ranks:
        x.reset(V0)
    pending, acc = None, []
    for e in range(args.epochs):
        for x in ranks:
            x.epoch()
        if (e + 1) % interval == 0:
            pending = reconcile(ranks, pending, rule)
        acc.append(accuracy(ranks[0].U, ranks[0].baseV if pending is not None else ranks[0].V,
                            ranks[0].baseB if pending is not None else ranks[0].B, probe))
    # the consolidated model: finish the last exchange
    reconcile(ranks, pending, rule)
    final = accuracy(ranks[0].U, ranks[0].baseV, ranks[0].baseB, probe)
    print("%-5s exchange every %d epoch(s) (%.1f updates per row and exchange): " % (rule, interval, interval * 2.0 * nnz / ni)
          + " ".join("%.3f" % a for a in acc[3::4]) + "   consolidated %.3f" % final, flush=True)
