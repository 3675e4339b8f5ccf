#!/usr/bin/env python3
"""Sweep of the binned hogwild BPR path against the fused atomic kernel at the ML-20M shape (k = 64 by default).
An arm is a dash-separated spec: `fused` (the default kernel), `share` (fused + shared negatives, bit 4; every other arm sets hogwild_flags bit 6), `wgN` (workgroups per CU of the triplet kernel),
`cNm` / `cNk` (chunk length), `hN` (hot threshold), `ablN` (profiling switches, hogwild_flags bits 8..).
Prints one line per arm: ms/epoch (HIP events around the kernels), triplets/s, the 'correct' fraction of the last
epoch and the pairwise loss / accuracy on a fixed probe sample after the same number of epochs."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cornac_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--epochs", type=int, default=6)
ap.add_argument("--lr", type=float, default=0.05)
ap.add_argument("--report-every", type=int, default=0)
ap.add_argument("--arms", default="fused,wg2,wg1")
args = ap.parse_args()
n_users, n_items, indptr, indices = bench.load_dataset("ml20m", 0, os.environ.get("TMPDIR", "/tmp"))
nnz = len(indices)
k = args.k
user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
rs = np.random.RandomState(1)
pick = rs.randint(nnz, size=400000)
pu, pi, pj = user_ids[pick], indices[pick], rs.randint(n_items, size=400000)


def probe(U, V, B):
    x = B[pi] - B[pj] + np.einsum("nk,nk->n", U[pu], V[pi] - V[pj])
    return float(np.mean(np.log1p(np.exp(-x)))), float(np.mean(x > 0))


def parse(spec):
    flags, env = 64, {}
    for tok in spec.split("-"):
        if tok == "fused":
            flags &= ~64
        elif tok == "share":      # fused kernel with negatives shared by groups of 4 sampling lanes (hogwild_flags bit 4)
            flags = (flags & ~64) | 16
        elif tok.startswith("wg"):
            env["CORNAC_HIP_BIN_WG_PER_CU"] = tok[2:]
        elif tok.startswith("c"):
            env["CORNAC_HIP_BIN_CHUNK"] = str(int(float(tok[1:-1]) * {"m": 1 << 20, "k": 1 << 10}[tok[-1]]))
        elif tok.startswith("h"):
            env["CORNAC_HIP_BIN_HOT"] = tok[1:]
        elif tok.startswith("ua"):
            env["CORNAC_HIP_BIN_UNRA"] = tok[2:]
        elif tok.startswith("ub"):
            env["CORNAC_HIP_BIN_UNRB"] = tok[2:]
        elif tok.startswith("abl"):
            flags |= int(tok[3:]) << 8
        else:
            raise SystemExit("bad arm token %r" % tok)
    return flags, env


for name in args.arms.split(","):
    flags, env = parse(name)
    for key in ("CORNAC_HIP_BIN_WG_PER_CU", "CORNAC_HIP_BIN_CHUNK", "CORNAC_HIP_BIN_HOT", "CORNAC_HIP_BIN_UNRA", "CORNAC_HIP_BIN_UNRB"):
        os.environ.pop(key, None)
    os.environ.update(env)
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    U, V, B = bench.init_factors(n_users, n_items, k, 100)
    tr.set_factors(U, V, B)
    tr.seed_hogwild(0xC0FFEE)
    tr.fit_epochs(1, args.lr, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags)
    tr.kernel_timing(True)
    t0 = time.perf_counter()
    c = s = 0
    for e in range(args.epochs):
        c, s = tr.fit_epochs(1, args.lr, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags)
        if args.report_every and (e + 2) % args.report_every == 0:
            U2, V2, B2 = tr.get_factors()
            print("   %-16s epochs %3d correct %.4f probe (loss, acc) %s maxV %.3f" % (name, e + 2, c / max(nnz - s, 1),
                  probe(U2, V2, B2), np.abs(V2).max()), flush=True)
    dt = time.perf_counter() - t0
    kms, launches = tr.kernel_timing(False)
    U2, V2, B2 = tr.get_factors()
    tr.close()
    loss, acc = probe(U2, V2, B2)
    print("%-16s flags %5d  kernel ms/epoch %7.3f  wall ms/epoch %7.3f  %.3f G triplets/s  launches/epoch %d  correct %.4f "
          "skipped %.4f  probe loss %.4f acc %.4f  finite %s" % (name, flags, kms / args.epochs, 1e3 * dt / args.epochs,
          nnz * args.epochs / dt / 1e9, launches // args.epochs, c / max(nnz - s, 1), s / nnz, loss, acc,
          bool(np.isfinite(V2).all() and np.isfinite(U2).all())), flush=True)
