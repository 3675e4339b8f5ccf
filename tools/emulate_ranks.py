#!/usr/bin/env python3
"""Convergence check of multi-GPU regime 1 WITHOUT several GPUs: R virtual ranks (each its own ML-20M-shaped user
population, as in the weak-scaling bench) run one after the other on one device with exactly the reconciliation
algebra of cornac_amd.dist.ItemTableReplica — every rank trains a chunk from the same base item table, the base
advances by the SUM of the ranks' deltas.  (The overlapped exchange of the real driver delays remote deltas by one
more chunk; --delay 1 emulates that.)  Prints the pairwise training accuracy per epoch next to a single rank's."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--epochs", type=int, default=6)
ap.add_argument("--syncs", type=int, default=16)
ap.add_argument("--delay", type=int, default=1)
ap.add_argument("--scale", type=float, default=0.25, help="fraction of the ML-20M shape per rank (memory/time)")
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--rule", default=None, help="avg | sum | sqrt (divide by sqrt of the touching ranks) | align")
ap.add_argument("--grid", default="", help='several configurations on one set of trainers, e.g. "sqrt:4,8,16;align:2,4,8,16"')
ap.add_argument("--sum", dest="avg", action="store_false", help="plain summation of the deltas (the rule that diverges) instead of the "
                "product's rule: summed delta of a row / number of ranks that touched it")
args = ap.parse_args()
n_users, n_items, nnz, a, seed = synth.CONFIGS["ml20m"]
n_users = int(n_users * args.scale); nnz = int(nnz * args.scale)
k, lr, reg = args.k, 0.05, 0.01
rs = np.random.RandomState(0)
V0 = ((rs.uniform(0, 1, (n_items, k)) - .5) / k).astype(np.float32)
trainers = []
for r in range(args.ranks):
    users, items = synth.zipf_interactions(n_users, n_items, nnz, a, seed + r)
    indptr, indices = synth.csr_from_sorted(users, items, n_users)
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    tr.set_factors(((rs.uniform(0, 1, (n_users, k)) - .5) / k).astype(np.float32), V0, np.zeros(n_items, np.float32))
    tr.seed_hogwild(1000 + r)
    trainers.append((tr, len(indices)))
    if r == 0:  # fixed probe triplets of rank 0's data: accuracy of the CONSOLIDATED item table after each epoch
        prs = np.random.RandomState(99)
        pp = prs.randint(0, len(indices), 200_000)
        probe_u = np.repeat(np.arange(n_users), np.diff(indptr))[pp]
        probe_i = indices[pp]
        probe_j = prs.randint(0, n_items, len(pp))
U_init = [tr.get_factors()[0] for tr, _ in trainers]


def run(rule, syncs):
    """one configuration on the shared trainers (tables and sample streams reset): prints per-epoch lines"""
    for r, (tr, _) in enumerate(trainers):
        tr.set_factors(U_init[r], V0, np.zeros(n_items, np.float32))
        tr.seed_hogwild(1000 + r)
    V, B = V0.copy(), np.zeros(n_items, np.float32)
    pending = []  # remote deltas not yet applied (overlap emulation)
    last = None
    for e in range(args.epochs):
        tot_c = tot_n = 0
        for c in range(syncs):
            dV, dB = np.zeros_like(V), np.zeros_like(B)
            cV, cB = np.zeros(n_items, np.float32), np.zeros(n_items, np.float32)
            qV, qB = np.zeros(n_items, np.float64), np.zeros(n_items, np.float64)
            for tr, m in trainers:
                tr.set_factors(None, V, B)
                n = m // syncs
                tr.hogwild_enqueue(n, lr, reg, True)
                cc, ss = tr.sync()
                Vr, Br = tr.get_item_factors()
                dV += Vr - V; dB += Br - B
                cV += (np.abs(Vr - V).max(1) > 0); cB += ((Br - B) != 0)
                qV += ((Vr - V).astype(np.float64) ** 2).sum(1); qB += (Br - B).astype(np.float64) ** 2
                tot_c += cc; tot_n += n - ss
            if rule == "avg":
                dV /= np.maximum(cV, 1)[:, None]; dB /= np.maximum(cB, 1)
            elif rule == "sqrt":
                dV /= np.sqrt(np.maximum(cV, 1))[:, None]; dB /= np.sqrt(np.maximum(cB, 1))
            elif rule == "align":  # S min(1, sum |d_r|^2 / |S|^2): cornac_amd.dist.ItemTableReplica rule="align"
                nV, nB = (dV.astype(np.float64) ** 2).sum(1), dB.astype(np.float64) ** 2
                dV *= np.where(nV > 0, np.minimum(1.0, qV / np.maximum(nV, 1e-300)), 1.0).astype(np.float32)[:, None]
                dB *= np.where(nB > 0, np.minimum(1.0, qB / np.maximum(nB, 1e-300)), 1.0).astype(np.float32)
            pending.append((dV, dB))
            if len(pending) > args.delay:
                d = pending.pop(0)
                V = V + d[0]; B = B + d[1]
        U0 = trainers[0][0].get_user_factors()
        sc = np.einsum("nk,nk->n", U0[probe_u], V[probe_i] - V[probe_j]) + B[probe_i] - B[probe_j]
        last = (tot_c / tot_n, float((sc > 0).mean()), float(np.abs(V).max()), bool(np.isfinite(V).all()))
        if not args.grid:
            print("epoch %d: pairwise accuracy while training %.4f | consolidated table on rank 0's probe triplets %.4f | |V| max %.3f finite %s"
                  % ((e,) + last), flush=True)
    return last


if args.grid:   # e.g. "sqrt:4,8,16;align:2,4,8,16": one line per configuration (the last epoch)
    for part in args.grid.split(";"):
        rule, counts = part.split(":")
        for syncs in [int(x) for x in counts.split(",")]:
            acc, cons, vmax, fin = run(rule, syncs)
            print("R = %d, rule %-5s %2d exchanges per epoch, %d epochs: accuracy while training %.4f | consolidated %.4f | max|V| %.3f finite %s"
                  % (args.ranks, rule, syncs, args.epochs, acc, cons, vmax, fin), flush=True)
else:
    run(args.rule or ("sqrt" if args.avg else "sum"), args.syncs)
