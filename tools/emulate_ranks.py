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
ap.add_argument("--rule", default=None, help="avg | sum | sqrt (divide by sqrt of the touching ranks)")
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
V, B = V0.copy(), np.zeros(n_items, np.float32)
pending = []  # remote deltas not yet applied (overlap emulation)
for e in range(args.epochs):
    tot_c = tot_n = 0
    for c in range(args.syncs):
        dV, dB = np.zeros_like(V), np.zeros_like(B)
        cV, cB = np.zeros(n_items, np.float32), np.zeros(n_items, np.float32)
        for tr, m in trainers:
            tr.set_factors(None, V, B)
            n = m // args.syncs
            tr.hogwild_enqueue(n, lr, reg, True)
            cc, ss = tr.sync()
            _, Vr, Br = tr.get_factors()
            dV += Vr - V; dB += Br - B
            cV += (np.abs(Vr - V).max(1) > 0); cB += ((Br - B) != 0)
            tot_c += cc; tot_n += n - ss
        rule = args.rule or ("sqrt" if args.avg else "sum")  # the product's rule is sqrt
        if rule == "avg":
            dV /= np.maximum(cV, 1)[:, None]; dB /= np.maximum(cB, 1)
        elif rule == "sqrt":
            dV /= np.sqrt(np.maximum(cV, 1))[:, None]; dB /= np.sqrt(np.maximum(cB, 1))
        pending.append((dV, dB))
        if len(pending) > args.delay:
            d = pending.pop(0)
            V = V + d[0]; B = B + d[1]
    U0 = trainers[0][0].get_factors()[0]
    sc = np.einsum("nk,nk->n", U0[probe_u], V[probe_i] - V[probe_j]) + B[probe_i] - B[probe_j]
    print("epoch %d: pairwise accuracy while training %.4f | consolidated table on rank 0's probe triplets %.4f | |V| max %.3f finite %s"
          % (e, tot_c / tot_n, float((sc > 0).mean()), np.abs(V).max(), np.isfinite(V).all()))
