#!/usr/bin/env python3
"""A/B of the XCD-strata BPR epoch (csrc/bpr_strata.inc) against the fused atomic kernel at the ML-20M shape.

An arm is a dash-separated spec: `atomic` (the fused kernel), `strata`, `ldsbin` (the form, hogwild_flags bits 16..19), then `uN` (LDS-bin triplets in flight, profile build), `xN` (LDS-bin hot_x1000), `sgN` / `hcN` (LDS-bin deal: strata_groups, hot_cost_x16),
`hN` (hot_permille), `mN` (hot_min_mult_x100), `rN` (rehash period), `vN` (kernel variant, profile build),
`ablN` (ablation bits, profile build).  Run with CORNAC_HIP_PROFILE=1 for the v / abl tokens.
Per arm: ms per epoch by HIP events (sum of the epoch's launches) and by wall clock, the 'correct' fraction of the last
epoch, the pairwise loss / accuracy on a fixed probe sample at the report epochs, and — from a separate 2-epoch run with
reg = 0 — the relative drift of V's column sums (0 for lossless updates: every triplet's item deltas cancel)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cornac_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--config", default="ml20m")
ap.add_argument("--epochs", type=int, default=20)
ap.add_argument("--report", default="5,10,20")
ap.add_argument("--lr", type=float, default=0.05)
ap.add_argument("--reg", type=float, default=0.01)
ap.add_argument("--arms", default="atomic,strata")
ap.add_argument("--cpu-threads", type=int, default=32, help="0 = no CPU hogwild reference")
args = ap.parse_args()
n_users, n_items, indptr, indices = bench.load_dataset(args.config, 0, os.environ.get("TMPDIR", "/tmp"))
nnz = len(indices)
k = args.k
user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
rs = np.random.RandomState(1)
pick = rs.randint(nnz, size=400000)
pu, pi, pj = user_ids[pick], indices[pick], rs.randint(n_items, size=400000)
report = sorted(int(x) for x in args.report.split(",") if x)


def probe(U, V, B):
    x = B[pi] - B[pj] + np.einsum("nk,nk->n", U[pu], V[pi] - V[pj])
    return float(np.mean(np.log1p(np.exp(-x)))), float(np.mean(x > 0))


def parse(spec):
    flags, cfg, env = 0, {}, {}
    for tok in spec.split("-"):
        if tok == "atomic":
            flags |= _lib.FORM_FUSED
        elif tok == "strata":
            flags |= _lib.FORM_STRATA
        elif tok == "ldsbin":
            flags |= _lib.FORM_LDSBIN
        elif tok.startswith("u"):
            env["CORNAC_HIP_LDSBIN_UNR"] = tok[1:]
        elif tok.startswith("x"):
            cfg["hot_x1000"] = int(tok[1:])
        elif tok.startswith("sg"):
            cfg["strata_groups"] = int(tok[2:])
        elif tok.startswith("hc"):
            cfg["hot_cost_x16"] = int(tok[2:])
        elif tok.startswith("abl"):
            flags |= int(tok[3:]) << 8
        elif tok.startswith("h"):
            cfg["hot_permille"] = int(tok[1:])
        elif tok.startswith("m"):
            cfg["hot_min_mult_x100"] = int(tok[1:])
        elif tok.startswith("r"):
            cfg["rehash_period"] = int(tok[1:])
        elif tok.startswith("v"):
            env["CORNAC_HIP_STRATA_VARIANT"] = tok[1:]
        else:
            raise SystemExit("bad arm token %r" % tok)
    return flags, cfg, env


def make(flags, cfg, env):
    os.environ.pop("CORNAC_HIP_STRATA_VARIANT", None)
    os.environ.pop("CORNAC_HIP_LDSBIN_UNR", None)
    os.environ.update(env)
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    cfg = dict(cfg)
    if "hot_x1000" in cfg:
        tr.ldsbin_config(hot_x1000=cfg.pop("hot_x1000"))
    if "strata_groups" in cfg or "hot_cost_x16" in cfg:
        tr.ldsbin_deal_config(strata_groups=cfg.pop("strata_groups", 16), hot_cost_x16=cfg.pop("hot_cost_x16", 32))
    if cfg:
        tr.strata_config(**cfg)
    U, V, B = bench.init_factors(n_users, n_items, k, 100)
    tr.set_factors(U, V, B)
    tr.seed_hogwild(0xC0FFEE)
    return tr, (U, V, B)


if args.cpu_threads:
    from oracle import oracle

    Uc, Vc, Bc = bench.init_factors(n_users, n_items, k, 100)
    done = 0
    threads = min(args.cpu_threads, oracle.lib().oracle_num_threads())
    for e in report:
        t0 = time.perf_counter()
        c, s = oracle.bpr_hogwild_epochs(indptr, indices, user_ids, n_items, Uc, Vc, Bc, k, args.lr, args.reg, True, 5 + e,
                                         threads, e - done)
        done = e
        print("cpu-%dthr            epochs %3d  (%.1f s)  probe (loss, acc) %s" % (threads, e, time.perf_counter() - t0,
              probe(Uc, Vc, Bc)), flush=True)

for name in args.arms.split(","):
    flags, cfg, env = parse(name)
    tr, _ = make(flags, cfg, env)
    tr.fit_epochs(1, args.lr, args.reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags)  # epoch 1 (also the warm-up)
    tr.kernel_timing(True)
    t0 = time.perf_counter()
    c = s = 0
    lines = []
    for e in range(2, args.epochs + 1):
        c, s = tr.fit_epochs(1, args.lr, args.reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags)
        if e in report:
            dt = time.perf_counter() - t0
            U2, V2, B2 = tr.get_factors()
            lines.append("e%d %s" % (e, "(%.4f, %.4f)" % probe(U2, V2, B2)))
            t0 += time.perf_counter() - t0 - dt  # do not bill the download + probe
    dt = time.perf_counter() - t0
    kms, launches = tr.kernel_timing(False)
    st = tr.strata_stats()
    lb = tr.ldsbin_stats() if (flags >> 16) in (0, 3) else None
    if lb and lb["bins"]:
        _, cold, off, _, _ = tr.debug_ldsbin_deal(0xC0FFEE, 3)
        tot = cold.astype(np.int64) + np.diff(off.astype(np.int64))
        lb["draws_max_over_mean"] = round(float(tot.max() / tot.mean()), 4)
        lb["cold_max_over_mean"] = round(float(cold.max() / cold.mean()), 4)
    tr.close()
    n_ep = args.epochs - 1
    # lossless-ness: column sums of V under reg = 0
    tr, (U, V, B) = make(flags, cfg, env)
    tr.fit_epochs(2, args.lr, 0.0, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags)
    _, V2, B2 = tr.get_factors()
    tr.close()
    moved = np.abs(V2.astype(np.float64) - V).sum(0)
    drift = np.abs(V2.astype(np.float64).sum(0) - V.astype(np.float64).sum(0))
    print("%-20s kernel %.3f ms/epoch (%d launches, %.3f ms each)  wall %.3f ms/epoch = %.3f G triplets/s | correct %.4f "
          "skipped %.4f | probe %s | colsum drift/moved %.2e | hot %d misplaced %d builds %d | %s"
          % (name, kms / n_ep, launches // n_ep, kms / max(launches, 1), 1e3 * dt / n_ep, nnz * n_ep / dt / 1e9,
             c / max(nnz - s, 1), s / nnz, "  ".join(lines), float((drift / moved.max()).max()), st["n_hot"],
             st["misplaced_workgroups"], st["bucket_builds"], lb), flush=True)
