#!/usr/bin/env python3
"""How many item-side exchanges per epoch does multi-GPU MF (cornac_amd.dist.ShardedMfTrainer) need?  CPU emulation, no
GPU: R virtual ranks, each with its own users' ratings of one shared catalogue, train slices of their epoch with the
ORACLE's fit_sgd loop (test infrastructure used as a simulator here, not as a product path) on their own replica of
[V | Bi]; the replicas are reconciled with exactly ItemTableReplica's algebra — delta of a row summed over the ranks /
sqrt(ranks that touched it), remote deltas arriving one slice late as in the overlapped exchange.  Reported per setting:
RMSE of the consolidated model on held-out ratings of every rank, next to (a) ONE process training on all the ratings and
(b) one rank alone on its own ratings.

    python tools/emulate_ranks_mf.py --ranks 8 --parts 1,2,4,8,16
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--parts", default="1,2,4,8,16")
ap.add_argument("--epochs", type=int, default=20)
ap.add_argument("--users-per-rank", type=int, default=1500)
ap.add_argument("--items", type=int, default=1700)
ap.add_argument("--per-user", type=int, default=60)
ap.add_argument("--k", type=int, default=16)
ap.add_argument("--lr", type=float, default=0.01)
ap.add_argument("--reg", type=float, default=0.02)
ap.add_argument("--rule", default="sqrt", help="sqrt | sum | avg | align | pX (divide by c^(X/100), e.g. p75)")
args = ap.parse_args()
R, ni, k = args.ranks, args.items, args.k


def make_rank(r):
    """ratings of rank r's users: shared item structure (low rank + bias, Zipf popularity), own user tastes"""
    gi = np.random.RandomState(5)
    q, ib = gi.normal(0, 1, (ni, 4)), gi.normal(0, 0.5, ni)
    p = 1.0 / np.arange(1, ni + 1) ** 0.8
    p /= p.sum()
    rs = np.random.RandomState(100 + r)
    rid, cid, val = [], [], []
    for u in range(args.users_per_rank):
        items = np.sort(rs.choice(ni, args.per_user, replace=False, p=p))
        t = rs.normal(0, 1, 4)
        rid.append(np.full(args.per_user, u)); cid.append(items)
        val.append(np.clip(3.0 + ib[items] + 0.5 * q[items] @ t + rs.normal(0, 0.4, args.per_user), 1, 5))
    rid, cid, val = np.concatenate(rid).astype(np.int64), np.concatenate(cid).astype(np.int64), np.concatenate(val).astype(np.float32)
    test = rs.rand(len(val)) < 0.1
    return (rid[~test], cid[~test], val[~test]), (rid[test], cid[test], val[test])


data = [make_rank(r) for r in range(R)]
mu = float(np.concatenate([d[0][2] for d in data]).astype(np.float64).mean())
init = np.random.RandomState(7)
V0, B0 = init.normal(0, 0.01, (ni, k)).astype(np.float32), np.zeros(ni, np.float32)


def new_users(seed):
    return np.random.RandomState(seed).normal(0, 0.01, (args.users_per_rank, k)).astype(np.float32), np.zeros(args.users_per_rank, np.float32)


def train_slice(tr, U, Bu, V, Bi, s0, s1):
    rid, cid, val = tr
    if s1 > s0:
        loss = np.zeros(1, np.float32)
        orc.lib().oracle_mf_fit(rid[s0:s1].copy(), cid[s0:s1].copy(), val[s0:s1].copy(), s1 - s0, U, V, Bu, Bi, k, args.lr,
                                args.reg, mu, 1, 1, 1, 0, loss.ctypes.data)


def rmse(U, Bu, V, Bi, te):
    rid, cid, val = te
    return float(np.sqrt(np.mean((mu + Bu[rid] + Bi[cid] + np.einsum("nk,nk->n", U[rid], V[cid]) - val) ** 2)))


def run(parts):
    users = [new_users(31 + r) for r in range(R)]
    flat = [(V0.copy(), B0.copy()) for _ in range(R)]
    base = (V0.copy(), B0.copy())
    pending = None   # (sum of deltas / rule, local deltas per rank) of the previous slice, applied one slice late
    for _ in range(args.epochs):
        for part in range(parts):
            for r in range(R):
                n = len(data[r][0][2])
                train_slice(data[r][0], users[r][0], users[r][1], flat[r][0], flat[r][1], n * part // parts, n * (part + 1) // parts)
            # begin_sync of this slice: local deltas and touch counts
            dV = [flat[r][0] - base[0] for r in range(R)]
            dB = [flat[r][1] - base[1] for r in range(R)]
            if pending is not None:   # finish_sync of the PREVIOUS exchange arrives now
                RV, RB, pV, pB = pending
                for r in range(R):
                    flat[r][0][...] += RV - pV[r]
                    flat[r][1][...] += RB - pB[r]
                base = (base[0] + RV, base[1] + RB)
                dV = [flat[r][0] - base[0] for r in range(R)]
                dB = [flat[r][1] - base[1] for r in range(R)]
            cV = sum((np.abs(d).max(1) > 0).astype(np.float32) for d in dV)
            cB = sum((d != 0).astype(np.float32) for d in dB)
            SV, SB = sum(dV), sum(dB)
            if args.rule == "sqrt":
                SV, SB = SV / np.sqrt(np.maximum(cV, 1))[:, None], SB / np.sqrt(np.maximum(cB, 1))
            elif args.rule == "avg":
                SV, SB = SV / np.maximum(cV, 1)[:, None], SB / np.maximum(cB, 1)
            elif args.rule == "align":
                # Delta = S * min(1, sum_r |d_r|^2 / |S|^2): the plain sum when the ranks' deltas of a row are orthogonal,
                # their mean when they are R copies of one step (ItemTableReplica rule="align")
                QV, QB = sum((d * d).sum(1) for d in dV), sum(d * d for d in dB)
                nV, nB = (SV * SV).sum(1), SB * SB
                SV = SV * np.minimum(1.0, QV / np.maximum(nV, 1e-30))[:, None]
                SB = SB * np.minimum(1.0, QB / np.maximum(nB, 1e-30))
            elif args.rule.startswith("p"):
                a = float(args.rule[1:]) / 100.0
                SV, SB = SV / (np.maximum(cV, 1) ** a)[:, None], SB / np.maximum(cB, 1) ** a
            pending = (SV, SB, dV, dB)
    RV, RB, pV, pB = pending
    V, Bi = base[0] + RV, base[1] + RB
    return float(np.mean([rmse(users[r][0], users[r][1], V, Bi, data[r][1]) for r in range(R)])), float(np.abs(V).max())


# (a) one process, all ratings (users of rank r get ids r * users_per_rank + u), same number of epochs
allU = np.concatenate([new_users(31 + r)[0] for r in range(R)])
allBu = np.zeros(len(allU), np.float32)
V, Bi = V0.copy(), B0.copy()
rid = np.concatenate([data[r][0][0] + r * args.users_per_rank for r in range(R)])
cid = np.concatenate([data[r][0][1] for r in range(R)])
val = np.concatenate([data[r][0][2] for r in range(R)])
for _ in range(args.epochs):
    train_slice((rid, cid, val), allU, allBu, V, Bi, 0, len(val))
one = float(np.mean([rmse(allU[r * args.users_per_rank:(r + 1) * args.users_per_rank], allBu[r * args.users_per_rank:(r + 1) * args.users_per_rank],
                          V, Bi, data[r][1]) for r in range(R)]))
# (b) rank 0 alone on its own ratings
U, Bu = new_users(31)
V, Bi = V0.copy(), B0.copy()
for _ in range(args.epochs):
    train_slice(data[0][0], U, Bu, V, Bi, 0, len(data[0][0][2]))
alone = rmse(U, Bu, V, Bi, data[0][1])
print("R = %d ranks x %d users x %d ratings, %d items, k = %d, %d epochs, rule %s | held-out RMSE: one process on all ratings %.4f, "
      "rank 0 alone %.4f" % (R, args.users_per_rank, args.per_user, ni, k, args.epochs, args.rule, one, alone), flush=True)
for parts in [int(x) for x in args.parts.split(",")]:
    e, vmax = run(parts)
    print("  %2d exchanges per epoch: consolidated held-out RMSE %.4f  (max|V| %.2f)" % (parts, e, vmax), flush=True)
