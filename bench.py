#!/usr/bin/env python3
"""Headline benchmark: BPR training throughput (triplets/s) + batched rank() throughput (items
scored/s) on an ML-20M-shaped synthetic interaction set, k = 64, fp32 tables (BASELINE.json
configs[1]), hogwild (throughput) mode — the counterpart of the reference's OpenMP path.

    python bench.py --gpus N --steps K --warmup W

A "step" is one BPR epoch (nnz = 20 000 263 sampled triplets) over the resident interaction
matrix.  For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU;
every rank owns a disjoint ML-20M-shaped user population (weak scaling) and the replicated item
table is reconciled by RCCL all-reduce of its deltas (cornac_amd/dist.py).

A plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) re-executes itself under
torch.distributed.run with N ranks on 127.0.0.1, so the one command works with and without an external launcher.

At N = 1 the line also carries `legs`: the other single-GPU configurations of BASELINE.json, each with its own
`roofline` and reference `cpu_baseline` — `mf_netflix` / `wmf_netflix` (configs[2]: biased MF and WMF, k = 128, at the
Netflix Prize shape),
`vbpr_tradesy` (configs[3]) and `bpr_k128_scale` (one GPU's user slice of configs[4], k = 128).

Rank 0 prints ONE JSON line (see README / DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3  # v_mfma_f32_32x32x2_f32 dense peak


def algorithmic_bytes_per_triplet(k, mean_degree):
    """3 rows + 2 biases read and written (24k + 16), user_ids/item_ids (8),
    indptr pair (8), ceil(log2(d+1)) 4-byte membership probes."""
    probes = int(np.ceil(np.log2(mean_degree + 1)))
    return 24 * k + 16 + 8 + 8 + 4 * probes, 8 + 8 + 4 * probes  # (processed triplet, skipped draw)


def load_dataset(name, seed_shift, cache_dir):
    from cornac_amd import synth

    n_users, n_items, nnz, a, seed = synth.CONFIGS[name]
    tag = "%s_s%d" % (name, seed + seed_shift)
    path = os.path.join(cache_dir, "cornac_amd_%s.npz" % tag)
    if os.path.exists(path):
        z = np.load(path)
        return n_users, n_items, z["indptr"], z["indices"]
    users, items = synth.zipf_interactions(n_users, n_items, nnz, a, seed + seed_shift)
    indptr, indices = synth.csr_from_sorted(users, items, n_users)
    try:
        np.savez(path, indptr=indptr, indices=indices)
    except OSError:
        pass
    return n_users, n_items, indptr, indices


SCALE = {"users_per_gpu": 12_500_000, "items": 10_000_000, "degree": 5, "k": 128}


def scale_slice(rank):
    """One GPU's user slice of BASELINE configs[4] (100 M users x 10 M items over 8 GPUs): 12.5 M users with 5 distinct
    items each, generated from seed 45 + rank.  Returns (n_users, n_items, indptr, indices)."""
    nu, ni, d = SCALE["users_per_gpu"], SCALE["items"], SCALE["degree"]
    rs = np.random.RandomState(45 + rank)
    base = rs.randint(0, ni, size=nu, dtype=np.int64)
    step = rs.randint(1, ni // (2 * d), size=nu, dtype=np.int64)
    items = (base[:, None] + step[:, None] * np.arange(d, dtype=np.int64)[None, :]) % ni
    items.sort(axis=1)
    indices = items.astype(np.int32).ravel()
    indptr = (np.arange(nu + 1, dtype=np.int64) * d).astype(np.int32)
    return nu, ni, indptr, indices


def scale_slice_zipf(rank, zipf_a=0.55, users=None):
    """The same slice with a popularity head: every user's 5 items drawn from Zipf(zipf_a) over a random permutation of the
    10 M items (SURVEY.md 8d C2's exponent; the most popular item then holds 0.026 % of the interactions — 23 x a passing
    bin's share), duplicates inside a user dropped.  Returns (n_users, n_items, indptr, indices)."""
    nu, ni, d = users or SCALE["users_per_gpu"], SCALE["items"], SCALE["degree"]
    rs = np.random.RandomState(4500 + rank)
    perm = np.random.RandomState(4499).permutation(ni).astype(np.int32)        # the same popularity on every rank
    items = np.empty((nu, d), np.int32)
    e = 1.0 - zipf_a
    for c0 in range(0, nu, 2_500_000):          # inverse CDF of the continuous power law x^-a on [1, ni + 1): rank = floor(x) - 1
        c1 = min(nu, c0 + 2_500_000)
        x = (rs.random_sample((c1 - c0, d)) * ((ni + 1.0) ** e - 1.0) + 1.0) ** (1.0 / e)
        items[c0:c1] = perm[np.minimum(x.astype(np.int64) - 1, ni - 1)]
    items.sort(axis=1)
    keep = np.ones(items.shape, bool)
    keep[:, 1:] = items[:, 1:] != items[:, :-1]
    indptr = np.zeros(nu + 1, np.int64)
    np.cumsum(keep.sum(1), out=indptr[1:])
    return nu, ni, indptr.astype(np.int32), items[keep]


def scale_factors(nu, ni, k, rank):
    """item table from a fixed seed (identical on every rank), user rows from a per-rank seed"""
    r2 = np.random.RandomState(2)
    V = ((r2.uniform(0, 1, (ni, k)).astype(np.float32) - 0.5) / k)
    ru = r2 if rank == 0 else np.random.RandomState(200 + rank)
    U = ((ru.uniform(0, 1, (nu, k)).astype(np.float32) - 0.5) / k)
    return U, V, np.zeros(ni, np.float32)


def init_factors(n_users, n_items, k, seed):
    rng = np.random.RandomState(seed)
    U = ((rng.uniform(0, 1, (n_users, k)).astype(np.float32) - 0.5) / k)
    V = ((rng.uniform(0, 1, (n_items, k)).astype(np.float32) - 0.5) / k)
    return U, V, np.zeros(n_items, np.float32)


def cpu_baseline_reference(indptr, indices, n_items, k, lr, reg, budget_s):
    """The REAL reference kernel: `BPR._fit_sgd` + `RNGVector` of the compiled extension in oracle/_ref (built from
    the sources under /root/reference by oracle/build_ref.py; loaded over the stand-in Python modules of
    oracle/ref_stubs where the reference tree itself is absent), driven with raw arrays exactly as `BPR.fit` drives
    it (cornac/models/bpr/recom_bpr.pyx:186-201).  One call = one epoch = nnz draws."""
    from oracle import ref_loader

    RNGVector, RefBPR = ref_loader.load_kernel_only()
    n_users = len(indptr) - 1
    nnz = len(indices)
    user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
    item_ids = np.ascontiguousarray(indices, np.int32)
    neg_item_ids = np.arange(n_items, dtype=np.int32)
    ip = np.ascontiguousarray(indptr, np.int32)
    U, V, B = init_factors(n_users, n_items, k, 1)
    model = RefBPR(k=k, learning_rate=lr, lambda_reg=reg)
    max_threads = os.cpu_count() or 1

    def epoch(t, seed):
        t0 = time.time()
        model._fit_sgd(RNGVector(t, nnz - 1, seed), RNGVector(t, n_items - 1, seed + 1), t, user_ids, item_ids,
                       neg_item_ids, ip, U, V, B)
        return time.time() - t0

    epoch(max_threads, 1)  # page in
    cands = sorted({t for t in (8, 16, 32, 64, max_threads) if t <= max_threads} | {max_threads})
    best = min((epoch(t, 10 + t), t) for t in cands)
    threads = best[1]
    n_epochs = int(max(1, min(8, round(budget_s / max(best[0], 1e-3)))))
    dt = sum(epoch(threads, 100 + e) for e in range(n_epochs))
    n = float(nnz) * n_epochs
    return {"value": n / dt, "unit": "triplets/s", "cores": threads, "kind": "reference",
            "sample": "%d epochs (%d triplets) of the same ML-20M-shaped matrix through the reference's compiled "
                      "BPR._fit_sgd (Cython/OpenMP, k=%d), %d threads (best of %s on a %d-thread host), %.1f s"
                      % (n_epochs, int(n), k, threads, cands, max_threads, dt)}


def cpu_rank_baseline(U, V, B, topk, n_sample=200):
    """The reference's per-user scoring + ranking flow on the host, timed over a sample of users: BPR.score
    (copy of the item biases + fast_dot, cornac/models/bpr/recom_bpr.pyx:272-297) followed by Recommender.rank's
    argpartition / argsort (cornac/models/recommender.py:503-530).  fast_dot is the reference's compiled extension
    when oracle/_ref is present, otherwise the C restatement in oracle/cornac_oracle.c."""
    kind = "port"
    fast_dot = None
    try:
        from oracle import ref_loader

        if ref_loader.kernel_available():
            ref_loader.load_kernel_only()
            import importlib

            fast_dot = importlib.import_module("cornac.utils.fast_dot").fast_dot
            kind = "reference fast_dot + numpy ranking as in Recommender.rank"
    except Exception:
        fast_dot = None
    if fast_dot is None:
        from oracle import oracle as orc

        fast_dot = orc.fast_dot
    n_items = V.shape[0]
    users = np.random.RandomState(0).randint(0, U.shape[0], n_sample)

    def one(u, k):
        scores = B.copy()
        fast_dot(U[u], V, scores)
        if k != -1:
            part = np.argpartition(scores, -k)
            top = part[-k:]
            part[-k:] = top[np.argsort(scores[top])]
            return part[::-1]
        return scores.argsort()[::-1]

    out = {}
    for name, k, n in (("topk", topk, n_sample), ("full", -1, max(10, n_sample // 4))):
        t_w = time.time()
        while time.time() - t_w < 0.5:  # the OpenMP team of fast_dot needs many calls to settle
            one(int(users[0]), k)
        t0 = time.time()
        for u in users[:n]:
            one(int(u), k)
        dt = time.time() - t0
        out[name] = {"ms_per_user": 1e3 * dt / n, "items_per_s": n * n_items / dt, "users_timed": int(n)}
    out["kind"] = kind
    out["cores"] = os.cpu_count() or 1
    return out


def cpu_baseline(indptr, indices, n_items, k, lr, reg, budget_s):
    """The reference's OpenMP Hogwild path timed on this host's cores over a bounded sample of the same workload:
    the real compiled reference kernel when oracle/_ref is present (kind "reference"), otherwise its restatement in
    oracle/cornac_oracle.c (same loop, same boost sampler, compiled with the reference's flags; kind "port")."""
    try:
        from oracle import ref_loader

        if ref_loader.kernel_available():
            return cpu_baseline_reference(indptr, indices, n_items, k, lr, reg, budget_s)
    except Exception as e:  # fall back to the port, but say why
        print("[bench] reference kernel unavailable (%s): timing the port" % e, file=sys.stderr)
    from oracle import oracle as orc

    L = orc.lib()
    max_threads = max(1, min(os.cpu_count() or 1, L.oracle_num_threads()))
    n_users = len(indptr) - 1
    user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
    U, V, B = init_factors(n_users, n_items, k, 1)
    nnz = len(indices)
    probe = min(nnz, 2_000_000)
    # Hogwild on many cores can be slower than on few (coherence traffic on hot rows): like a user
    # tuning num_threads, probe a few thread counts and time the best one.
    cands = sorted({t for t in (8, 16, 32, 64, max_threads) if t <= max_threads} | {max_threads})
    orc.bpr_hogwild_epochs(indptr, indices, user_ids, n_items, U, V, B, k, lr, reg, True, 4, max_threads, 1,
                           num_samples=probe)  # page in
    best = (0.0, max_threads)
    for t in cands:
        t0 = time.time()
        orc.bpr_hogwild_epochs(indptr, indices, user_ids, n_items, U, V, B, k, lr, reg, True, 5, t, 1,
                               num_samples=probe)
        best = max(best, (probe / (time.time() - t0), t))
    rate, threads = best
    n = int(min(nnz * 4, max(probe, rate * budget_s)))
    t0 = time.time()
    orc.bpr_hogwild_epochs(indptr, indices, user_ids, n_items, U, V, B, k, lr, reg, True, 6, threads, 1,
                           num_samples=n)
    dt = time.time() - t0
    return {"value": n / dt, "unit": "triplets/s", "cores": threads, "kind": "port",
            "sample": "%d BPR triplets (%.2f epoch) of the same ML-20M-shaped matrix, k=%d, OpenMP hogwild, "
                      "%d threads (best of %s on a %d-thread host), %.1f s" % (n, n / nnz, k, threads, cands, max_threads, dt)}


# ======================================================================================================================
# extra single-GPU legs (N = 1): BASELINE.json configs[2], [3] and one GPU's slice of [4]
# ======================================================================================================================
def synth_ratings(n_users, n_items, nnz, zipf_a, seed):
    """COO ratings sorted by user at a dataset's SHAPE, generated in O(nnz) without a global de-duplication (100 M
    entries in seconds): per-user counts ~ multinomial(log-normal activity), items ~ Zipf over a random permutation,
    ratings 1..5 around user + item offsets.  (MF trains on the COO list as given; duplicates are harmless.)"""
    rs = np.random.RandomState(seed)
    act = rs.lognormal(0.0, 1.0, n_users)
    counts = rs.multinomial(nnz, act / act.sum())
    users = np.repeat(np.arange(n_users, dtype=np.int64), counts)
    p_item = 1.0 / np.arange(1, n_items + 1) ** zipf_a
    cdf = np.cumsum(p_item / p_item.sum())
    perm = rs.permutation(n_items).astype(np.int64)
    items = np.empty(nnz, np.int64)
    step = 1 << 24
    for a in range(0, nnz, step):   # chunked: bounds the float64 temporaries
        b = min(a + step, nnz)
        items[a:b] = perm[np.searchsorted(cdf, rs.random_sample(b - a)).clip(0, n_items - 1)]
    bu, bi = rs.normal(0, 0.5, n_users).astype(np.float32), rs.normal(0, 0.5, n_items).astype(np.float32)
    val = np.empty(nnz, np.float32)
    for a in range(0, nnz, step):
        b = min(a + step, nnz)
        val[a:b] = np.clip(np.rint(3.5 + bu[users[a:b]] + bi[items[a:b]] + rs.normal(0, 0.7, b - a)), 1, 5)
    return users, items, val


def leg_traffic(name, **must_match):
    """counter-measured bytes per launch of a leg's dominant kernel (profiles/traffic.json, taken by tools/pmc_legs.sh), or
    None when the file does not describe this kernel / workload"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["legs"][name]
    except Exception:
        return None
    for key, val in must_match.items():
        if tj.get(key) != val:
            print("[bench] profiles/traffic.json leg %s was measured with %s = %r (now %r): traffic = null"
                  % (name, key, tj.get(key), val), file=sys.stderr)
            return None
    return tj.get("bytes_per_launch")


def leg_mf_netflix(args, _lib):
    """configs[2]: biased MF, k = 128, Netflix Prize shape (480 189 x 17 770, 100 480 507 ratings), hogwild mode.
    Algorithmic bytes per rating: U and V rows read + written (16 k), the two biases R+W (16), the COO
    record (int64 rid, int64 cid, f32 val = 20).  Item popularity: Zipf exponent 0.45 (the most-rated title holds 0.25 %
    of the ratings, as in the real set: 232 944 of 100 480 507); SURVEY 8d's suggested 0.8 (2.8 % on one item row) is
    measured beside it (`zipf_0.8`)."""
    from cornac_amd import synth

    n_users, n_items, nnz, zipf_a, seed = synth.CONFIGS["netflix"]
    k, lr, reg = 128, 0.01, 0.02
    b = 16 * k + 16 + 20

    def run(a, epochs):
        t0 = time.time()
        users, items, val = synth_ratings(n_users, n_items, nnz, a, seed)
        t_gen = time.time() - t0
        rs = np.random.RandomState(1)
        mu = float(val.mean())
        t0 = time.time()
        tr = _lib.MfTrainer(users, items, val, n_users, n_items, k)
        U = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
        V = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
        tr.set_factors(U, V, np.zeros(n_users, np.float32), np.zeros(n_items, np.float32))
        tr.fit(1, lr, reg, mu, True, False, _lib.MODE_HOGWILD)  # warm-up: builds the ownership tables
        t_setup = time.time() - t0
        tr.kernel_timing(True)
        t0 = time.perf_counter()
        loss, _ = tr.fit(epochs, lr, reg, mu, True, False, _lib.MODE_HOGWILD)
        dt = time.perf_counter() - t0
        kms, launches = tr.kernel_timing(False)
        tr.close()
        # an epoch is one launch of the fused kernel or 8 phase launches of the block rotation (csrc/mf_blocks.inc)
        rotation = launches == 8 * epochs
        kernel = ("mf_blocks_kernel<2,4> (8 launches = one epoch)" if rotation else "mf_hogwild_rowwise_kernel (one launch = one epoch)")
        per_launch = nnz * epochs / max(launches, 1)
        achieved = per_launch * b / (kms / max(launches, 1) / 1e3) / 1e9
        top = float(np.bincount(items, minlength=n_items).max()) / nnz
        return dict(users=users, items=items, val=val, mu=mu, t_gen=t_gen, t_setup=t_setup, dt=dt, epochs=epochs, loss=loss,
                    kms=kms, launches=launches, rotation=rotation, kernel=kernel, per_launch=per_launch, achieved=achieved,
                    top=top)

    r = run(zipf_a, 3)
    out = {"metric": "mf_ratings_per_sec", "value": nnz * r["epochs"] / r["dt"], "unit": "ratings/s", "steps": r["epochs"],
           "ms_per_step": 1e3 * r["dt"] / r["epochs"], "dtype": "f32", "data": "synthetic",
           "config": {"workload": "biased MF k=%d, Netflix-Prize-shaped synthetic ratings (%d users x %d items, %d "
                                  "ratings, int64 COO as the reference's uir_tuple), item popularity Zipf exponent %.2f "
                                  "(hottest item row: %.2f %% of the ratings), hogwild mode"
                                  % (k, n_users, n_items, nnz, zipf_a, 100 * r["top"]),
                      "lr": lr, "reg": reg, "form": "block rotation" if r["rotation"] else "fused atomic kernel",
                      "zipf_exponent": zipf_a},
           "roofline": {"bound": "hbm", "achieved": r["achieved"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": r["achieved"] / HBM_PEAK_GBS,
                        "traffic": leg_traffic("mf_netflix", kernel=r["kernel"], ratings_per_launch=int(r["per_launch"]), k=int(k)),
                        "kernel": r["kernel"],
                        "launches": r["launches"], "avg_launch_ms": r["kms"] / max(r["launches"], 1),
                        "algorithmic_bytes_per_rating": b},
           "train_stats": {"mse_per_epoch": [float(x) / nnz for x in r["loss"]]},
           "host_s": {"generate": r["t_gen"], "create_and_first_epoch": r["t_setup"]}}
    if args.cpu_baseline_seconds > 0:
        out["cpu_baseline"] = cpu_baseline_mf(r["users"], r["items"], r["val"], n_users, n_items, k, lr, reg, r["mu"],
                                              args.cpu_baseline_seconds)
        if out["cpu_baseline"]:
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    del r
    if os.environ.get("CORNAC_BENCH_MF_ZIPF08", "1") != "0":
        try:
            z = run(0.8, 1)
            out["zipf_0.8"] = {"value": nnz * z["epochs"] / z["dt"], "unit": "ratings/s", "ms_per_step": 1e3 * z["dt"] / z["epochs"],
                               "hottest_item_row_share": z["top"], "form": "block rotation" if z["rotation"] else "fused atomic kernel",
                               "frac": z["achieved"] / HBM_PEAK_GBS, "frac_of_step": nnz * b / (z["dt"] / z["epochs"]) / 1e9 / HBM_PEAK_GBS,
                               "mse_per_epoch": [float(x) / nnz for x in z["loss"]],
                               "note": "the same shape with SURVEY 8d's Zipf exponent: one item row holds 3.2 % of the ratings; rows "
                                       "above 0.1 % are split into virtual rows (one per 0.1 %), trained like any hot row and merged "
                                       "after every phase with the align rule (csrc/mf_blocks.inc; round 4 kept such a row in one LDS "
                                       "bin under its lock: 679 ms per epoch)"}
        except Exception as e:
            out["zipf_0.8"] = {"error": repr(e)}
    return out


def cpu_baseline_mf(users, items, val, n_users, n_items, k, lr, reg, mu, budget_s):
    """The reference's `backend_cpu.fit_sgd` (cornac/models/mf/backend_cpu.pyx:35-97, compiled in oracle/_ref) on a
    bounded sample of the same ratings: one epoch = one pass over the sample's COO list."""
    try:
        from oracle import ref_loader

        ref_fit = ref_loader.load_mf_kernel()
        kind = "reference"

        def fit_sgd(rid, cid, v, U, V, Bu, Bi, iters, threads):
            ref_fit(rid, cid, v, U, V, Bu, Bi, lr, reg, mu, iters, threads, True, False, False)
    except Exception as e:
        print("[bench] reference MF kernel unavailable (%r): timing the port" % (e,), file=sys.stderr)
        try:
            from oracle import oracle as orc

            L, kind = orc.lib(), "port"

            def fit_sgd(rid, cid, v, U, V, Bu, Bi, iters, threads):
                loss = np.zeros(iters, np.float32)
                L.oracle_mf_fit(rid, cid, v, len(v), U, V, Bu, Bi, k, lr, reg, mu, iters, threads, 1, 0, loss)
        except Exception as e2:
            print("[bench] mf cpu_baseline failed: %r" % (e2,), file=sys.stderr)
            return None
    n = min(len(val), 20_000_000)
    rs = np.random.RandomState(3)
    rid, cid, v = np.ascontiguousarray(users[:n]), np.ascontiguousarray(items[:n]), np.ascontiguousarray(val[:n])
    nu = int(rid.max()) + 1
    U = rs.normal(0, 0.01, (nu, k)).astype(np.float32)
    V = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
    Bu, Bi = np.zeros(nu, np.float32), np.zeros(n_items, np.float32)
    max_threads = os.cpu_count() or 1

    def run(threads, iters):
        t0 = time.time()
        fit_sgd(rid, cid, v, U, V, Bu, Bi, iters, threads)
        return time.time() - t0

    run(max_threads, 1)  # page in
    cands = sorted({t for t in (16, 32, 64, max_threads) if t <= max_threads})
    best = min((run(t, 1), t) for t in cands)
    iters = int(max(1, min(6, round(budget_s / max(best[0], 1e-3)))))
    dt = run(best[1], iters)
    return {"value": n * iters / dt, "unit": "ratings/s", "cores": best[1], "kind": kind,
            "sample": "%d epochs over the first %d ratings (%d users) of the same COO list through the reference's compiled "
                      "backend_cpu.fit_sgd (Cython/OpenMP, k=%d), %d threads (best of %s on a %d-thread host), %.1f s"
                      % (iters, n, nu, k, best[1], cands, max_threads, dt)}


def leg_vbpr_tradesy(args, _lib):
    """configs[3]: VBPR k = k2 = 64 with 4096-d visual features at the Tradesy shape (19 243 users x 165 906 items,
    394 421 feedback), batch 100 as the reference's default.  The step is bound by torch.optim.Adam's dense sweep
    over every table (recom_vbpr.py:228-262): parameter and two moments read and written = 6 passes x 4 bytes over all
    parameters (rows outside the batch have an exactly zero gradient, which is not read) + the 2 B feature rows."""
    nu, ni, nnz, nf, k, k2, B = 19243, 165906, 394421, 4096, 64, 64, 100
    nnz = int(os.environ.get("CORNAC_BENCH_VBPR_FEEDBACK", nnz))  # (counter-collection runs shorten the epoch: tools/pmc_legs.sh)
    rs = np.random.RandomState(44)
    t0 = time.time()
    F = rs.random_sample((ni, nf)).astype(np.float32)
    u = rs.randint(0, nu, nnz).astype(np.int32)
    i = rs.randint(0, ni, nnz).astype(np.int32)
    j = rs.randint(0, ni, nnz).astype(np.int32)
    t_gen = time.time() - t0
    tr = _lib.VbprTrainer(F, nu, ni, k, k2)
    lim = np.sqrt(3.0) * np.sqrt(2.0 / (nu + k))
    params = dict(Bi=np.zeros(ni, np.float32), Gu=rs.uniform(-lim, lim, (nu, k)), Gi=rs.uniform(-lim, lim, (ni, k)),
                  Tu=rs.uniform(-lim, lim, (nu, k2)), E=rs.uniform(-0.03, 0.03, (nf, k2)), Bp=rs.uniform(-0.03, 0.03, nf))
    tr.set_params(**params)
    tr.fit_batches(u[:2000], i[:2000], j[:2000], B, 0.005, 0.01, 0.01, 0.0)  # warm-up
    t0 = time.perf_counter()
    nll = tr.fit_batches(u, i, j, B, 0.005, 0.01, 0.01, 0.0)
    dt = time.perf_counter() - t0
    tr.close()
    steps = (nnz + B - 1) // B
    n_par = ni + nu * k + ni * k + nu * k2 + nf * k2 + nf
    bytes_step = 24.0 * n_par + 2.0 * B * nf * 4
    out = {"metric": "vbpr_triplets_per_sec", "value": nnz / dt, "unit": "triplets/s", "steps": steps,
           "ms_per_step": 1e3 * dt / steps, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "VBPR k=k2=%d, %d-d features, Tradesy-shaped synthetic (%d users x %d items, %d "
                                  "feedback), batch %d, one epoch on pre-sampled batches" % (k, nf, nu, ni, nnz, B)},
           "roofline": {"bound": "hbm", "achieved": bytes_step * steps / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bytes_step * steps / dt / 1e9 / HBM_PEAK_GBS,
                        "traffic": leg_traffic("vbpr_tradesy", shape="%dx%d f%d k%d k2 %d batch%d" % (nu, ni, nf, k, k2, B)),
                        "kernel": "whole minibatch step (featdiff gather, projection GEMM, pair gradient, scatter, dense "
                                  "Adam sweep): bytes / wall time of the epoch, no per-kernel events",
                        "algorithmic_bytes_per_step": bytes_step},
           "train_stats": {"mean_nll_per_pair": nll / (steps * B * B)}, "host_s": {"generate": t_gen}}
    if args.cpu_baseline_seconds > 0:
        try:
            out["cpu_baseline"] = cpu_baseline_vbpr(F, u, i, j, params, nu, ni, k, k2, B, args.cpu_baseline_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:
            print("[bench] vbpr cpu_baseline failed: %r" % (e,), file=sys.stderr)
            out["cpu_baseline"] = None
    return out


def cpu_baseline_vbpr(F, u, i, j, params, nu, ni, k, k2, B, budget_s):
    """The reference's VBPR step is PyTorch on the host (recom_vbpr.py:228-262); oracle/vbpr_oracle.py restates it
    line by line (bit-identical to the live reference, tests/test_oracle_vs_reference.py) — timed over a few batches."""
    import torch

    from oracle import vbpr_oracle

    torch.set_num_threads(min(32, os.cpu_count() or 1))
    n_steps, dt = vbpr_oracle.timed_steps(F, params, u, i, j, B, budget_s=min(budget_s, 8.0))
    return {"value": n_steps * B / dt, "unit": "triplets/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d minibatch steps of %d triplets of the same tables through the torch restatement of the "
                      "reference's VBPR step (oracle/vbpr_oracle.py; the reference itself is torch on the host), "
                      "%d threads, %.1f s" % (n_steps, B, torch.get_num_threads(), dt)}


def _scale_fit(_lib, nu, ni, indptr, indices, k, epochs):
    """one handle over a configs[4]-sized slice: (c, skipped, seconds, kernel ms, launches, ldsbin stats, setup seconds)"""
    t0 = time.time()
    tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
    U, V, B = scale_factors(nu, ni, k, 0)
    tr.set_factors(U, V, B)
    del U, V, B
    tr.seed_hogwild(7)
    tr.fit_epochs(1, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)  # warm-up: builds the form's tables
    t_setup = time.time() - t0
    lb = tr.ldsbin_stats()
    tr.kernel_timing(True)
    t0 = time.perf_counter()
    c, sk = tr.fit_epochs(epochs, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    dt = time.perf_counter() - t0
    kms, launches = tr.kernel_timing(False)
    lb["lock_timeouts"] = tr.ldsbin_stats()["lock_timeouts"]
    tr.close()
    return c, sk, dt, kms, launches, lb, t_setup


def leg_bpr_k128_scale(args, _lib):
    """One GPU's share of configs[4] (100 M users x 10 M items, k = 128, 8 GPUs): users are partitioned, so a rank
    owns 12.5 M users; the item table (10 M x 128 fp32 = 5.1 GB) is held whole.  5 distinct items per user.  Timed over 8
    epochs (round 5: 2).  `zipf_0.55`: the same slice with a Zipf(0.55) popularity head (scale_slice_zipf)."""
    d, k = SCALE["degree"], SCALE["k"]
    t0 = time.time()
    nu, ni, indptr, indices = scale_slice(0)
    t_gen = time.time() - t0
    nnz = len(indices)
    epochs = 8
    c, sk, dt, kms, launches, lb, t_setup = _scale_fit(_lib, nu, ni, indptr, indices, k, epochs)
    b_full, b_skip = algorithmic_bytes_per_triplet(k, d)
    skip = sk / float(nnz * epochs)
    # an epoch is one launch (LDS-bin form with passing bins — the automatic choice when an epoch draws >= 2 interactions
    # per item row —, or the fused kernel) or 8 partition launches of the XCD-strata form: bytes per launch = the epoch's
    # bytes x epochs / launches recorded
    bytes_launch = nnz * ((1 - skip) * b_full + skip * b_skip) * epochs / max(launches, 1)
    strata = launches == 8 * epochs
    passing = lb["bins"] > 0 and not strata
    kernel = ("bpr_ldsbin_kernel<2,4> (passing bins: %d bins of <= %d item rows, %d threads, %d B of LDS each)"
              % (lb["bins"], lb["rows_per_bin"], lb["block_threads"], lb["lds_bytes"]) if passing
              else "bpr_strata_kernel<2,2>" if strata else "bpr_hogwild_rowwise_kernel<64,2,2,atomic,owned>")
    out = {"metric": "bpr_triplets_per_sec", "value": nnz * epochs / dt, "unit": "triplets/s", "steps": epochs,
           "ms_per_step": 1e3 * dt / epochs, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BPR k=%d on one GPU's user slice of the 100 M x 10 M synthetic (%d users x %d items, "
                                  "%d interactions, U %.1f GB, V %.1f GB: beyond the Infinity Cache), hogwild mode"
                                  % (k, nu, ni, nnz, nu * k * 4 / 1e9, ni * k * 4 / 1e9),
                      "form": "ldsbin (passing bins)" if passing else "strata" if strata else "fused",
                      "sampling": ("every draw picks an interaction with probability 1 / nnz (nnz draws per epoch); the "
                                   "negative is uniform over the ~%d items sharing the positive's LDS bin in that epoch, bins "
                                   "re-dealt every epoch (every item pair can meet: csrc/bpr_ldsbin.inc); item-row updates "
                                   "exact (LDS read-modify-write under a row lock), user rows by fp32 atomics"
                                   % lb["rows_per_bin"]) if passing else
                                  "XCD strata: the negative is uniform over the eighth of the items dealt to the positive's "
                                  "partition in that epoch; plain read-modify-write of item rows inside one XCD" if strata else
                                  "global uniform negatives, fp32 atomics"},
           # `frac` is over the whole step (the per-epoch bucket deal and the bias pad / unpad passes included): the epoch's
           # algorithmic bytes / ms_per_step; the 8 partition launches alone: frac_kernel_only
           "roofline": {"bound": "hbm", "achieved": bytes_launch * launches / dt / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bytes_launch * launches / dt / 1e9 / HBM_PEAK_GBS,
                        "traffic": leg_traffic("bpr_k128_scale", kernel="bpr_ldsbin_kernel<2,4> (passing bins)", k=int(k),
                                               draws_per_launch=int(nnz)) if passing else None,
                        "frac_kernel_only": bytes_launch / (kms / max(launches, 1) / 1e3) / 1e9 / HBM_PEAK_GBS,
                        "kernel": kernel, "launches": launches,
                        "avg_launch_ms": kms / max(launches, 1), "algorithmic_bytes_per_triplet": b_full},
           "train_stats": {"correct_frac": c / max(nnz * epochs - sk, 1), "skipped_frac": skip},
           "host_s": {"generate": t_gen, "create_and_first_epoch": t_setup}}
    if passing:
        # what bounds the passing-bin kernel: the item rows cross the memory system once per epoch (LDS), every processed
        # triplet still issues the k / 16 fp32 atomic requests (64 B granules) of its user row
        req = nnz * (1 - skip) * (k / 16.0) / (dt / epochs) / 1e9
        out["roofline"]["limiter"] = {"what": "memory-side fp32 atomic requests (64 B granules) of the user rows",
                                      "requests_per_triplet": k / 16.0, "achieved_G_per_s": req, "probe_ceiling_G_per_s": 20.0,
                                      "frac_of_probe_ceiling": req / 20.0,
                                      "evidence": "profiles/r05_scale_pmc.csv, profiles/r05_exp_scale_passing.log (ablations: no "
                                                  "updates 19.8 ms, user rows by racy read-modify-write instead of atomics 32.0 vs 33.8)"}
    if os.environ.get("CORNAC_BENCH_SCALE_ZIPF", "1") != "0":
        try:
            t0 = time.time()
            zu, zi, zip_, zix = scale_slice_zipf(0)
            zt = time.time() - t0
            zn = len(zix)
            zc, zs, zdt, zk, zl, zlb, _ = _scale_fit(_lib, zu, zi, zip_, zix, k, epochs)
            zskip = zs / float(zn * epochs)
            zb_full, zb_skip = algorithmic_bytes_per_triplet(k, zn / zu)
            zbytes = zn * ((1 - zskip) * zb_full + zskip * zb_skip)
            deg = np.bincount(zix, minlength=zi)
            out["zipf_0.55"] = {"value": zn * epochs / zdt, "ms_per_step": 1e3 * zdt / epochs, "steps": epochs,
                                "frac": zbytes * epochs / zdt / 1e9 / HBM_PEAK_GBS,
                                "frac_kernel_only": zbytes * epochs / (zk / 1e3) / 1e9 / HBM_PEAK_GBS, "launches": zl,
                                "interactions": zn, "top_item_share": float(deg.max()) / zn,
                                "bins": zlb["bins"], "rows_per_bin": zlb["rows_per_bin"], "n_hot": zlb["n_hot"],
                                "hot_interaction_share": zlb["hot_interactions"] / float(zn),
                                # a hot triplet side updates its item row by k / 16 memory-side atomic requests on top of the
                                # user row's: the hot rows' share of all atomic requests
                                "hot_row_atomic_share": zlb["hot_interactions"] / float(zn + zlb["hot_interactions"]),
                                "lock_timeouts": zlb["lock_timeouts"], "correct_frac": zc / max(zn * epochs - zs, 1),
                                "skipped_frac": zskip, "generate_s": zt,
                                "workload": "the same slice, every user's 5 items from Zipf(0.55) over the 10 M items"}
            del zip_, zix, deg
        except Exception as e:
            print("[bench] scale zipf variant failed: %r" % (e,), file=sys.stderr)
            out["zipf_0.55"] = {"error": repr(e)}
    # this leg's rate differs by up to 25 % between GPU boxes (11.5 GB of randomly accessed tables); the line carries a
    # calibration of the box it ran on — copy, streaming read and random 512-byte gather rates over 6 GiB buffers — and
    # the partition modes rocm-smi reports, so that a slow run can be told from a slow box
    try:
        out["box"] = _lib.device_probe(0, 6 << 30)
        import subprocess
        smi = subprocess.run(["rocm-smi", "--showmemorypartition", "--showcomputepartition"], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True, timeout=20).stdout
        out["box"]["partitions"] = [ln.strip() for ln in smi.splitlines() if "artition" in ln and "GPU[0]" in ln]
    except Exception as e:
        out["box"] = {"error": repr(e)}
    if args.cpu_baseline_seconds > 0:
        try:
            # the reference kernel on the first 1 M users of the same matrix (its U slice is 0.5 GB instead of 6.4 GB)
            sub = 1_000_000
            cb = cpu_baseline(indptr[:sub + 1], indices[:sub * d], ni, k, 0.05, 0.01, min(args.cpu_baseline_seconds, 8.0))
            cb["sample"] = "first %d users of the same matrix; " % sub + cb["sample"].replace("ML-20M-shaped", "scale-leg")
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        except Exception as e:
            print("[bench] scale cpu_baseline failed: %r" % (e,), file=sys.stderr)
            out["cpu_baseline"] = None
    return out


def leg_wmf_netflix(args, _lib):
    """configs[2], the WMF half: one Adam step = one batch of 128 items against ALL 480 189 users (wmf.py:34-55), k = 128.
    Bound: fp32 MFMA — three GEMM-shaped pieces of 2 n_users B k flops each per step."""
    import scipy.sparse as sp

    n_users, n_items, k, B = 480_189, 17_770, 128, 128
    nnz = 20_000_000  # the step cost depends on n_users, B, k and the batch's non-zeros only: a fifth of the ratings
    rs = np.random.RandomState(0)
    t0 = time.time()
    keys = np.unique(rs.randint(0, n_users * n_items, size=int(nnz * 1.05), dtype=np.int64))[:nnz]
    users, items = keys // n_items, keys % n_items
    R = sp.csc_matrix((rs.randint(1, 6, len(users)).astype(np.float32), (users, items)), shape=(n_users, n_items))
    t_gen = time.time() - t0
    tr = _lib.WmfTrainer(R, k)
    lim = np.sqrt(6.0 / (n_users + k))
    U0 = rs.uniform(-lim, lim, (n_users, k)).astype(np.float32)
    V0 = rs.uniform(-lim, lim, (n_items, k)).astype(np.float32)
    tr.set_factors(U0, V0)
    perm = rs.permutation(n_items)
    batches = [perm[a:a + B] for a in range(0, n_items - B + 1, B)][:60]
    tr.fit_batches(batches[:3], 0.01, 0.01, 1.0, 0.01, 0.001)  # warm-up: workspaces, kernel attributes
    tr.kernel_timing(True)
    # Two passes over the same 60 batches, the second one is the figure: a WMF fit runs thousands of steps back to back, and
    # the first tens of milliseconds after an idle device are slower (the rank leg's five calls: 5.0 -> 4.5 ms); the first
    # pass is reported beside it.
    tr.fit_batches(batches, 0.01, 0.01, 1.0, 0.01, 0.001)
    dev_ms_first = tr.last_device_ms()
    t0 = time.perf_counter()
    loss = tr.fit_batches(batches, 0.01, 0.01, 1.0, 0.01, 0.001)
    dt = time.perf_counter() - t0
    dev_ms = tr.last_device_ms()
    tr.close()
    steps = len(batches)
    flops = 6.0 * n_users * B * k
    out = {"metric": "wmf_steps_per_sec", "value": steps / dt, "unit": "Adam steps/s (128 items x all users each)",
           "steps": steps, "ms_per_step": 1e3 * dt / steps, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "WMF k=%d, %d users x %d items, batches of %d items, a=1 b=0.01 (one epoch = %d steps)"
                                  % (k, n_users, n_items, B, (n_items + B - 1) // B)},
           "roofline": {"bound": "mfma", "achieved": flops * steps / (dev_ms / 1e3) / 1e12, "peak": FP32_MFMA_PEAK_TF,
                        "unit": "TFLOP/s", "frac": flops * steps / (dev_ms / 1e3) / 1e12 / FP32_MFMA_PEAK_TF,
                        "kernel": "wmf_user_step_ws_kernel (4 MFMA waves + 4 streaming waves per CU) + gather / reduce / "
                                  "item-side Adam: HIP events around the whole batch loop", "flops_per_step": flops, "device_ms_per_step": dev_ms / steps, "device_ms_per_step_first_pass": dev_ms_first / steps,
                        "traffic": leg_traffic("wmf_netflix", n_users=n_users, k=k, batch=B)},
           "train_stats": {"loss_first_last": [float(loss[0]), float(loss[-1])]}, "host_s": {"generate": t_gen},
           "parity": "oracle pinned to the reference's own WMF code run over oracle/tf1_shim (no TensorFlow in the image: its "
                     "Adam / gather-gradient rules are restated)"}
    if args.cpu_baseline_seconds > 0:
        try:
            from oracle import wmf_oracle

            o = wmf_oracle.WmfOracle(U0, V0, R, 0.01, 0.01, 1.0, 0.01, 0.001)
            n, t0 = 0, time.time()
            while time.time() - t0 < min(args.cpu_baseline_seconds, 8.0) and n < len(batches):
                o.step(batches[n])
                n += 1
            dtc = time.time() - t0
            out["cpu_baseline"] = {"value": n / dtc, "unit": out["unit"], "cores": os.cpu_count() or 1, "kind": "port",
                                   "sample": "%d steps of the same batches through the numpy restatement of the reference's "
                                             "TensorFlow graph (oracle/wmf_oracle.py, BLAS threads as configured), %.1f s" % (n, dtc)}
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:
            print("[bench] wmf cpu_baseline failed: %r" % (e,), file=sys.stderr)
            out["cpu_baseline"] = None
    return out


def _one_rank_group():
    """a process group of ONE rank over RCCL (the multi-GPU drivers on the only GPU of the box); returns a closer"""
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return lambda: None
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    return dist.destroy_process_group


CONVEYOR_SAMPLING = ("every draw picks an interaction with probability 1 / nnz (nnz draws per rank and epoch, each by the rank that "
                     "owns its user); the negative is uniform over the ~%d items sharing the positive's LDS bin in that epoch; the "
                     "bins — and the conveyor's blocks, which are ranges of %d bins — are re-dealt from ALL items every epoch with a "
                     "key the ranks share (rows move to their new slots in one all_to_all at the epoch boundary), so every item "
                     "pair can meet (csrc/bpr_ldsbin.inc, tests/test_dist_cpu.py::test_conveyor_blocks_are_redealt...); item-row "
                     "updates exact (LDS read-modify-write under a row lock), user rows by fp32 atomics; no hot-item path")


def conveyor_one_rank(args, torch, dev, rings, epochs, plain_s=None, virtual_world=8):
    """the configs[4] slice through BinConveyorBprTrainer on ONE rank laid out like a node of `virtual_world` ranks: per-step launch,
    per-step block copy (communication stream), per-epoch re-deal — each timed with events on its own stream"""
    from cornac_amd.dist import BinConveyorBprTrainer

    nu, ni, indptr, indices = scale_slice(0)
    k = SCALE["k"]
    U, V, B = scale_factors(nu, ni, k, 0)
    t0 = time.time()
    ring = BinConveyorBprTrainer(indptr, indices, nu, ni, k, dev, seed=11, emulate_traffic=True, rings=rings, virtual_world=virtual_world)
    ring.set_user_factors(U)
    ring.load_items(V, B)
    del U, V, B
    ring.run_epoch(args.lr, args.reg)
    ring.finish()
    t_setup = time.time() - t0
    mem = torch.cuda.memory_allocated(dev)
    ring.timing = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        ring.run_epoch(args.lr, args.reg)
    c, s = ring.finish()
    torch.cuda.synchronize()
    driven = (time.perf_counter() - t0) / epochs
    ev = ring.timing_summary()
    st = ring.trainer.tr.ldsbin_stats()
    out = {"driver_ms_per_epoch": 1e3 * driven, "epochs_timed": epochs,
           "protocol": "conveyor laid out for %d ranks on one: %d blocks of %d bins on %d ring(s), %d steps per epoch, ONE launch per "
                       "step over %d bin range(s), %.0f MB copied per step on the communication stream, rows re-dealt every epoch"
                       % (virtual_world, ring.nb_total, ring.bpb, ring.K, ring.nb, ring.K, ring.K * ring.bufs[0][0].numel() * 4 / 1e6),
           "blocks": ring.nb_total, "bins": ring.n_bins, "rows_per_bin": ring.cap, "block_threads": st["block_threads"],
           "launch_ms": ev["launch"][1], "launches": ev["launch"][0], "move_ms": ev["move"][1], "redeal_ms": ev["redeal"][1], "redeal_ms_min": ev["redeal"][2],
           "redeals": ev["redeal"][0], "triplets_per_s_driver": ring.nnz / driven, "skipped_frac": s / float(ring.nnz * epochs),
           "correct_frac": c / max(ring.nnz * epochs - s, 1), "lock_timeouts": st["lock_timeouts"], "setup_s": t_setup,
           "torch_bytes_allocated": int(mem), "sampling": CONVEYOR_SAMPLING % (ring.cap, ring.bpb)}
    if plain_s is not None:
        out["plain_ms_per_epoch"] = 1e3 * plain_s
        out["tax"] = 1.0 - plain_s / driven
    ring.close()
    del ring
    torch.cuda.empty_cache()
    return out


def leg_dist_tax(args, _lib):
    """What the multi-GPU driver costs BEFORE a second GPU is involved: one rank through RCCL (process group of one, the
    replicated item table bound to the handle, delta passes, all-reduce, overlapped schedule) next to the plain
    fit_epochs call, same data, same tables, same kernel form, at the ML-20M shape and at the configs[4] slice.
    tax = 1 - plain time / driver time.  How often and by which rule the replicas are reconciled: cornac_amd.dist.exchange_schedule
    (ML-20M shape: 16 exchanges per epoch, "sqrt", from inside one launch per epoch; configs[4] slice, `scale`: one exchange per
    epoch, "align" — what regime 1 does there when asked; that shape's own regime is the conveyor, `scale_ring`)."""
    import torch

    from cornac_amd.dist import ShardedBprTrainer, exchange_schedule

    close = _one_rank_group()
    dev = torch.device("cuda", 0)
    out = {"metric": "one_rank_driver_tax", "unit": "fraction of the driver's time", "higher_is_better": False}
    try:
        for shape in [x for x in os.environ.get("CORNAC_BENCH_DIST_TAX_SHAPES", "ml20m,scale").split(",") if x]:
            if shape == "scale":
                nu, ni, indptr, indices = scale_slice(0)
                k, epochs = SCALE["k"], 8
                U, V, B = scale_factors(nu, ni, k, 0)
            else:
                nu, ni, indptr, indices = load_dataset("ml20m", 0, args.cache_dir)
                k, epochs = 64, 10
                U, V, B = init_factors(nu, ni, k, 100)
            nnz = len(indices)
            spe, interval, rule = exchange_schedule(nnz, ni)
            tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
            tr.set_factors(U, V, B)
            tr.seed_hogwild(11)
            tr.fit_epochs(1, args.lr, args.reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
            t0 = time.perf_counter()
            tr.fit_epochs(epochs, args.lr, args.reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
            plain = (time.perf_counter() - t0) / epochs
            tr.close()
            tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
            tr.set_factors(U, V, B)
            tr.seed_hogwild(11)
            sh = ShardedBprTrainer(tr, ni, k, dev, sync_every=(nnz + spe - 1) // spe, sparse_threshold=None, rule=rule)
            sh.load_items(V, B)
            timed = {}
            # the resident exchange (ONE launch per epoch, the exchange points inside it) where the handle takes the LDS-bin
            # form, chunk launches with the overlapped exchange between them otherwise: ShardedBprTrainer.run_epoch decides
            resident = os.environ.get("CORNAC_BENCH_DIST_CHUNKS") is None and 1 <= spe <= 32 and sh.resident_bins() > 0
            # twice: as RCCL runs it on one rank (a process group of one launches NO kernel for an all-reduce), and with
            # a stand-in where each all-reduce sits that streams what a ring all-reduce over 8 ranks moves through a rank
            # (2 x 7/8 x the bucket) with 16 workgroups — the collective's memory traffic and CU share, not its link time
            for world in (0, 8):
                sh.table.emulate_world = world
                for _ in range(2 * interval):   # warm-up with the timed region's own pattern (begin, step, finish: both buffer sets exist)
                    sh.run_epoch(nnz, spe, args.lr, args.reg, True, _lib.NEG_UNIFORM, 0, resident=resident, epochs_per_exchange=interval)
                sh.finish()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(epochs):
                    sh.run_epoch(nnz, spe, args.lr, args.reg, True, _lib.NEG_UNIFORM, 0, resident=resident, epochs_per_exchange=interval)
                sh.finish()
                torch.cuda.synchronize()
                timed[world] = (time.perf_counter() - t0) / epochs
            driven = timed[0]
            tr.close()
            del sh
            torch.cuda.empty_cache()
            out[shape] = {"plain_ms_per_epoch": 1e3 * plain, "driver_ms_per_epoch": 1e3 * driven,
                          "exchanges_per_epoch": spe if interval == 1 else 1.0 / interval, "rule": rule, "epochs_timed": epochs,
                          "tax": 1.0 - plain / driven,
                          "driver_ms_per_epoch_with_8_rank_standin": 1e3 * timed[8], "tax_with_8_rank_standin": 1.0 - plain / timed[8],
                          "protocol": "resident exchange (one launch per epoch)" if resident else "chunk launches",
                          "triplets_per_s_plain": nnz / plain, "triplets_per_s_driver": nnz / driven,
                          "workload": "%d users x %d items, %d interactions, k = %d" % (nu, ni, nnz, k)}
        # regime 2 at the configs[4] slice: the conveyor (cornac_amd.dist.BinConveyorBprTrainer) laid out for EIGHT ranks on this
        # one — 16 blocks (bin ranges of the epoch's deal), a step = one launch over a sixteenth of the bins, exactly a node
        # rank's launch size; the trained block is copied to the free buffer on the communication stream beside the next
        # step's launch (what a neighbour's receive writes on a node: table / 16 per step), and the rows are re-dealt to the
        # next epoch's slots at every epoch boundary (on a node: one all_to_all of table / 8 per rank)
        if "scale" in out and os.environ.get("CORNAC_BENCH_RING", "1") != "0":
            try:
                out["scale_ring"] = conveyor_one_rank(args, torch, dev, rings=1, epochs=4, plain_s=out["scale"]["plain_ms_per_epoch"] / 1e3)
                out["scale_ring"]["workload"] = out["scale"]["workload"]
            except Exception as e:  # (the other shapes of the leg stay in the line)
                print("[bench] dist_tax scale_ring failed: %r" % (e,), file=sys.stderr)
                out["scale_ring"] = {"error": repr(e)}
        out["value"] = max(v["tax"] for v in out.values() if isinstance(v, dict) and "tax" in v)
    finally:
        close()
    return out


LEGS = {"dist_tax": leg_dist_tax, "mf_netflix": leg_mf_netflix, "wmf_netflix": leg_wmf_netflix, "vbpr_tradesy": leg_vbpr_tradesy, "bpr_k128_scale": leg_bpr_k128_scale}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    port = os.environ.get("MASTER_PORT", str(29400 + os.getpid() % 500))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main_ring(args, _lib, torch, dist, dev, world, rank, k, barrier):
    """`--config scale` over a process group: configs[4] as the ring conveyor (cornac_amd.dist.BinConveyorBprTrainer — every rank
    its own 12.5 M users, the 10 M x k item table as 2 N K blocks = bin ranges of the epoch's deal that rotate over K strided
    rings; a bench step = one epoch = 2 N launches per rank, each beside the previous blocks' transfer, plus the re-deal of the
    rows at the boundary).  Same JSON contract; the roofline block prices the launches (HIP events on the compute stream)."""
    from cornac_amd.dist import BinConveyorBprTrainer

    n_users, n_items, indptr, indices = scale_slice(rank)
    nnz = len(indices)
    U, V, B = scale_factors(n_users, n_items, k, rank)
    t0 = time.time()
    # the popularity order of the WHOLE matrix: every rank must deal the same items to the same bins
    from cornac_amd.dist import global_item_degrees

    order = np.argsort(-global_item_degrees(indices, n_items, dev, None), kind="stable").astype(np.int32)
    ring = BinConveyorBprTrainer(indptr, indices, n_users, n_items, k, dev, seed=0xC0FFEE, emulate_traffic=(world == 1), rings=args.rings,
                                 item_order=order, virtual_world=(args.virtual_world or None) if world == 1 else None)
    ring.set_user_factors(U)
    ring.load_items(V, B)
    del U, V, B
    t_setup = time.time() - t0
    for _ in range(args.warmup):
        ring.run_epoch(args.lr, args.reg)
    ring.finish()
    ring.timing = True
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ring.run_epoch(args.lr, args.reg)
    correct, skipped = ring.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ev = ring.timing_summary()
    st = ring.trainer.tr.ldsbin_stats()
    if rank == 0:
        b_full, b_skip = algorithmic_bytes_per_triplet(k, nnz / n_users)
        skip = skipped / float(nnz * args.steps)
        step_bytes = nnz * ((1 - skip) * b_full + skip * b_skip)                        # rank 0's share of one epoch
        gbs = step_bytes * args.steps / elapsed / 1e9
        launch_s = ev["launch"][1] / 1e3
        out = {"metric": "bpr_triplets_per_sec", "value": float(nnz) * args.steps * world / elapsed, "unit": "triplets/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BPR k=%d on one GPU's user slice per rank of the 100 M x 10 M synthetic (%d users x %d items, "
                                      "%d interactions per GPU), hogwild mode, fp32 tables resident in HBM"
                                      % (k, n_users, n_items, nnz), "k": k, "lr": args.lr, "reg": args.reg,
                          "form": "ldsbin (passing bins), conveyor layout: %d bins of <= %d rows" % (ring.n_bins, ring.cap),
                          "sampling": CONVEYOR_SAMPLING % (ring.cap, ring.bpb),
                          "parallelism": "user-partitioned dp%d, item table sharded by row into %d blocks (bin ranges of the epoch's "
                                         "deal) on %d ring(s) (regime 2, BinConveyorBprTrainer): an epoch = %d steps per rank, a step = "
                                         "ONE launch over %d bin range(s) beside the transfer of the previously trained block(s) "
                                         "(%.0f MB each) to the next rank%s; rows re-dealt at every epoch boundary"
                                         % (world, ring.nb_total, ring.K, ring.nb, ring.K, ring.bufs[0][0].numel() * 4 / 1e6,
                                            " — one rank: the block is copied on the communication stream instead" if world == 1 else "")},
               "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "traffic": None, "kernel": "bpr_ldsbin_kernel<2,4> over a block's bins (frac: the whole epoch's bytes / "
                                                       "wall time, transfers and the re-deal included)",
                            "frac_kernel_only": (step_bytes / ring.nb / launch_s / 1e9 / HBM_PEAK_GBS) if launch_s else None,
                            "avg_launch_ms": ev["launch"][1], "launches": ev["launch"][0],
                            "algorithmic_bytes_per_triplet": b_full},
               "per_step": {"launch_ms": ev["launch"][1], "transfer_ms": ev["move"][1], "redeal_ms_per_epoch": ev["redeal"][1], "redeal_ms_min": ev["redeal"][2],
                            "steps_per_epoch": ring.nb, "transfer_MB": ring.K * ring.bufs[0][0].numel() * 4 / 1e6},
               "train_stats": {"correct_frac": correct / max(nnz * args.steps - skipped, 1.0), "skipped_frac": skip,
                               "lock_timeouts": st["lock_timeouts"]},
               "host_s": {"setup": t_setup}, "cpu_baseline": None}
        print(json.dumps(out))
    ring.close()
    dist.destroy_process_group()


def dry_run(args):
    """The launcher / rendezvous / timing / JSON scaffolding with a stand-in step and the gloo backend: what the CPU
    test of `--gpus N` exercises (tests/test_bench_launcher_cpu.py).  No HIP code runs."""
    import torch
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))  # uneven ranks: the reported time must be the slowest one's
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "bpr_triplets_per_sec", "value": 1000.0 * args.steps * world / elapsed,
                          "unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "dry-run",
                          "config": {"workload": "dry run of the launcher (no GPU work)", "parallelism": "dp%d" % world}}))


SAMPLING = {
    "ldsbin": "every draw picks an interaction with the reference's probability 1 / nnz (nnz draws per epoch); the negative is "
              "uniform over the ~n_items / bins items dealt to the positive's LDS bin in that epoch instead of uniform over all "
              "items; bins are re-dealt every epoch from popularity strata permuted by an epoch-keyed bijection, so every bin "
              "carries the same popularity mass and ANY two items share a bin with probability ~1 / bins per epoch (no pair is "
              "excluded: tests/test_ldsbin_deal_cpu.py); item updates exact (LDS read-modify-write under a row lock), user rows "
              "and hot item rows by fp32 atomics",
    "strata": "stratified by user ownership (every wave draws its positives i.i.d. from its own users' interactions) and by XCD "
              "item partition (8 partitions re-dealt every epoch; the negative is uniform inside the positive's partition); "
              "item rows by plain read-modify-write inside one XCD (racy like the reference's threads)",
    "fused": "stratified by user ownership: every wave of the persistent grid draws its positives i.i.d. from its own users' "
             "interactions (len(slice) draws per epoch), negatives uniform over all items - not the reference's single global "
             "draw stream; every item-row update a device-scope fp32 atomic",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="ml20m", help="ml20m (BASELINE configs[1], the headline) | scale (configs[4]: every "
                    "rank holds one GPU's 12.5 M-user slice of the 100 M x 10 M synthetic, k = 128)")
    ap.add_argument("--k", type=int, default=0, help="0 = the config's own (64; scale: 128)")
    ap.add_argument("--sparse-threshold", type=float, default=-1.0,
                    help="replicated-table regime: exchange (row id, delta row) records instead of the dense table when every "
                         "rank touched at most this fraction of the item rows since the last exchange; < 0 = the config's own "
                         "(ml20m: dense always; scale: 0.5)")
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--reg", type=float, default=0.01)
    ap.add_argument("--flags", type=int, default=0, help="hogwild_flags of cornac_hip_bpr_fit_epochs")
    ap.add_argument("--sync-per-epoch", type=int, default=0,
                    help="item-table exchanges per epoch (N > 1); 0 = cornac_amd.dist.exchanges_per_epoch of the rank's "
                         "interaction count (16 at the ML-20M shape, 1 at the configs[4] slice)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--no-rank", action="store_true")
    ap.add_argument("--rank-users", type=int, default=0, help="users ranked in the scoring leg (0 = all)")
    ap.add_argument("--rank-full-users", type=int, default=10000, help="users of the full-ranking (k = -1) probe")
    ap.add_argument("--cache-dir", default=os.environ.get("TMPDIR", "/tmp"))
    ap.add_argument("--sharded-items", action="store_true",
                    help="multi-GPU regime 2: item table sharded by row, all-to-all of the touched rows "
                         "(default for N > 1 is regime 1: replicated item table + all-reduce of deltas)")
    ap.add_argument("--micro-batch", type=int, default=2_000_000, help="draws per exchange with --sharded-items")
    ap.add_argument("--dist-chunks", action="store_true",
                    help="multi-GPU regime 1: cut the epoch into chunk launches with the overlapped exchange between them "
                         "even where the resident exchange (one launch per epoch) is available")
    ap.add_argument("--force-dist", action="store_true",
                    help="exercise the multi-GPU code path (process group, bound item table, all-reduce) with 1 rank")
    ap.add_argument("--virtual-world", type=int, default=0,
                    help="--config scale on ONE rank: lay the conveyor out for this many ranks (a node rank's launch size)")
    ap.add_argument("--rings", type=int, default=1,
                    help="--config scale over ranks: the conveyor's item blocks ride this many strided rings at once (different "
                         "xGMI links; 4 at N = 8), each moving a K-th of a step's bytes")
    ap.add_argument("--replicated-items", action="store_true",
                    help="--config scale over N > 1 ranks: regime 1 (replicated item table, delta all-reduce at exchange_schedule's "
                         "interval) instead of the default for that shape, regime 2 as a ring conveyor of item blocks")
    ap.add_argument("--legs", default="mf_netflix,wmf_netflix,vbpr_tradesy,bpr_k128_scale,dist_tax",
                    help="extra single-GPU legs reported under `legs` at N = 1 (comma list; empty = none)")
    ap.add_argument("--no-legs", action="store_true")
    ap.add_argument("--dry-run-cpu", action="store_true", help="launcher / timing scaffolding only, gloo, no GPU")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    if args.dry_run_cpu:
        return dry_run(args)

    import torch

    from cornac_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libcornac_hip has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=dev)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    scale = args.config == "scale"
    k = args.k or (SCALE["k"] if scale else 64)
    if scale and distributed and not args.replicated_items and not args.sharded_items:
        return main_ring(args, _lib, torch, dist, dev, world, rank, k, barrier)
    if scale:
        n_users, n_items, indptr, indices = scale_slice(rank)
        args.no_rank = args.no_legs = True  # the scoring leg and the other legs belong to the headline configuration
    else:
        n_users, n_items, indptr, indices = load_dataset(args.config, rank, args.cache_dir)
    nnz = len(indices)
    trainer = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k, device=local_rank)
    if scale:
        U, V, B = scale_factors(n_users, n_items, k, rank)
    else:
        U, V, B = init_factors(n_users, n_items, k, 100 + rank)
        if distributed:
            V, B = init_factors(n_users, n_items, k, 100)[1:]  # identical item table on every rank
    trainer.set_factors(U, V, B)
    trainer.seed_hogwild(0xC0FFEE + 7919 * rank)
    # the form a whole-epoch call with these flags takes (include/cornac_hip.h: hogwild_flags bits 16..19)
    trainer_stats = {"ldsbin": trainer.ldsbin_stats()}
    sel = (args.flags >> 16) & 15
    # (profile builds honour the ablation bits 8..15 inside the LDS-bin / strata forms; the shipped library takes the fused
    # kernel for any non-zero low flag)
    low = args.flags & (0xff if os.environ.get("CORNAC_HIP_PROFILE") == "1" else 0xffff)
    form = ("fused" if low or sel == 1 or (distributed and args.sharded_items) else
            "ldsbin" if sel in (0, 3) and trainer_stats["ldsbin"]["bins"] > 0 else
            "strata" if sel == 2 or (sel == 0 and n_items >= 1 << 20) else "fused")

    sharded, resident_mode, exchange_interval, exchange_rule = None, False, 1, "sqrt"
    if distributed and args.sharded_items:
        from cornac_amd.dist import RowShardedBprTrainer

        sharded = RowShardedBprTrainer(trainer, n_items, k, dev, micro_batch=args.micro_batch)
        sharded.load_items(V, B)
    elif distributed:
        from cornac_amd.dist import ShardedBprTrainer

        from cornac_amd.dist import exchange_schedule

        # how often the replicas are reconciled and by which rule: cornac_amd.dist.exchange_schedule (16 per epoch under
        # "sqrt" at the ML-20M shape; one exchange every 4 epochs under "align" at the configs[4] slice)
        exchange_interval, exchange_rule = 1, ("sqrt" if args.sync_per_epoch >= 16 else "align")
        if args.sync_per_epoch <= 0:
            args.sync_per_epoch, exchange_interval, exchange_rule = exchange_schedule(nnz, n_items)
        # (sparse records pay off when a rank touches a small part of the table per exchange; with one exchange per epoch
        # at the configs[4] density every row is touched: dense, which also keeps the fused finish + begin pass)
        sparse = args.sparse_threshold if args.sparse_threshold >= 0 else (0.5 if scale and args.sync_per_epoch > 4 else None)
        if sparse is not None:
            exchange_interval = 1
        sharded = ShardedBprTrainer(trainer, n_items, k, dev, sync_every=(nnz + args.sync_per_epoch - 1)
                                    // args.sync_per_epoch, sparse_threshold=sparse, rule=exchange_rule)
        sharded.load_items(V, B)
        resident_mode = (not args.dist_chunks and sparse is None and 1 <= args.sync_per_epoch <= 32
                         and sharded.resident_bins(_lib.NEG_UNIFORM, args.flags) > 0)

    def step():
        if sharded is None:
            return trainer.fit_epochs(1, args.lr, args.reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, args.flags)
        if args.sharded_items:
            sharded.run(nnz, args.lr, args.reg, True)
        else:
            # the resident exchange (one launch per epoch, the exchange points inside it) where the handle takes the LDS-bin
            # form; chunk launches with the overlapped exchange between them otherwise or with --dist-chunks
            sharded.run_epoch(nnz, args.sync_per_epoch, args.lr, args.reg, True, _lib.NEG_UNIFORM, args.flags,
                              resident=resident_mode, epochs_per_exchange=exchange_interval)
        return (0, 0)

    for _ in range(args.warmup):
        step()
    if sharded is not None:
        sharded.finish()
    trainer.kernel_timing(enable=True)  # start recording HIP events around the SGD kernel launches
    barrier()
    t0 = time.perf_counter()
    correct = skipped = 0
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        c, s = step()   # (single GPU: returns after the epoch's counters have been fetched, i.e. after the kernel)
        step_ms.append(1e3 * (time.perf_counter() - ts))
        correct += c
        skipped += s
    if sharded is not None:
        correct, skipped = sharded.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = trainer.kernel_timing(enable=False)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_triplets = float(nnz) * args.steps * world
    value = total_triplets / elapsed
    out = {
        "metric": "bpr_triplets_per_sec", "value": value, "unit": "triplets/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BPR k=%d on one GPU's user slice per rank of the 100 M x 10 M synthetic (%d users x %d items, "
                                "%d interactions per GPU, U %.1f GB + V %.1f GB per GPU), hogwild mode, fp32 tables resident in HBM"
                                % (k, n_users, n_items, nnz, n_users * k * 4 / 1e9, n_items * k * 4 / 1e9)) if scale else
                               "BPR k=%d on ML-20M-shaped synthetic interactions (%d users x %d items, nnz %d per "
                               "GPU), hogwild mode, fp32 tables resident in HBM" % (k, n_users, n_items, nnz),
                   "k": k, "lr": args.lr, "reg": args.reg, "hogwild_flags": args.flags,
                   "form": form,
                   "sampling": SAMPLING[form],
                   "parallelism": "1 gpu" if world == 1 and not distributed else
                                  ("user-partitioned dp%d, item table sharded by row, all-to-all every %d draws"
                                   % (world, args.micro_batch)) if args.sharded_items else
                                  ("user-partitioned dp%d, item table all-reduce %s, rule %s%s"
                                   % (world, "x%d/epoch" % args.sync_per_epoch if exchange_interval == 1 else
                                      "every %d epochs" % exchange_interval, exchange_rule,
                                      " from inside one launch per epoch (resident exchange)"
                                      if resident_mode else ", chunk launches"))},
    }
    if rank == 0 and sharded is None:
        out["step_ms"] = {"min": float(np.min(step_ms)), "median": float(np.median(step_ms)), "max": float(np.max(step_ms))}
    if rank == 0:
        mean_deg = nnz / n_users
        b_full, b_skip = algorithmic_bytes_per_triplet(k, mean_deg)
        n_draws = float(nnz) * args.steps
        skip_frac = skipped / n_draws if n_draws else 0.0
        draws_per_launch = n_draws / max(launches, 1)  # rank 0's launches (an epoch is split in sync chunks for N > 1)
        bytes_per_launch = draws_per_launch * ((1.0 - skip_frac) * b_full + skip_frac * b_skip)
        avg_launch_s = (kernel_ms / 1e3) / max(launches, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if launches else None
        kernel_name = ("sample/apply kernels of the row-sharded path (no fused SGD kernel)" if args.sharded_items else
                       "bpr_ldsbin_kernel<2,4> (passing bins)" if form == "ldsbin" and trainer_stats["ldsbin"]["block_threads"] == 512 and 64 < k <= 128 else
                       "bpr_ldsbin_kernel<%d,%d>" % ((k + 63) // 64, {1: 4, 2: 2, 3: 2, 4: 1}[(k + 63) // 64]) if form == "ldsbin" else
                       "bpr_strata_kernel<%d,%d>" % ((k + 63) // 64, {1: 4, 2: 2, 3: 2, 4: 1}[(k + 63) // 64]) if form == "strata" else
                       "bpr_hogwild_rowwise_kernel<64,%d,%d,atomic,owned>" % ((k + 63) // 64, {1: 4, 2: 2, 3: 2, 4: 1}[(k + 63) // 64]))
        # counter-measured traffic is only quoted for the kernel + workload it was taken on (profiles/traffic.json
        # names both); any other launch configuration reports null rather than a stale number
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                same = (tj.get("kernel") == kernel_name and tj.get("hogwild_flags") == args.flags and
                        tj.get("k") == k and tj.get("config") == args.config and tj.get("draws_per_launch") == int(draws_per_launch))
                traffic = tj.get("bpr_hogwild_bytes_per_launch") if same else None
                if not same:
                    print("[bench] profiles/traffic.json was measured on another kernel / workload: traffic = null",
                          file=sys.stderr)
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                           "kernel": kernel_name, "launches": launches,
                           "avg_launch_ms": 1e3 * avg_launch_s, "algorithmic_bytes_per_triplet": b_full,
                           "skip_fraction": skip_frac}
        if achieved and form == "ldsbin":
            # what bounds the LDS-bin kernel: item rows never leave the LDS, every processed triplet still
            # issues the 64-byte fp32 atomic requests of its user row (k / 16) plus those of the hot item rows
            lb = trainer_stats["ldsbin"]
            # `frac` prices a triplet with SURVEY 8d's 24k + 48 bytes; this kernel moves neither the indptr pair nor the
            # log2(d) CSR probes of that figure (one user id, one item slot, one bitmap word instead): the same rate by the
            # bytes the kernel itself needs
            own_full, own_skip = 24 * k + 16 + 4 + 4 + 4, 4 + 4 + 4
            out["roofline"]["frac_by_the_kernels_own_bytes"] = (out["roofline"]["frac"] * ((1.0 - skip_frac) * own_full + skip_frac * own_skip)
                                                                / ((1.0 - skip_frac) * b_full + skip_frac * b_skip))
            out["roofline"]["own_bytes_per_triplet"] = own_full
            req_per = k / 16.0 + (lb["hot_interactions"] / float(nnz)) * (k / 16.0 + 1.0)
            req = (n_draws - skipped) / max(launches, 1) * req_per / avg_launch_s / 1e9
            out["roofline"]["limiter"] = {"what": "memory-side fp32 atomic requests (64 B granules) of the user rows and the hot item rows",
                                          "requests_per_triplet": req_per, "achieved_G_per_s": req, "probe_ceiling_G_per_s": 20.0,
                                          "frac_of_probe_ceiling": req / 20.0, "bins": lb["bins"], "rows_per_bin": lb["rows_per_bin"],
                                          "hot_items": lb["n_hot"], "evidence": "profiles/r05_sgd_pmc.csv (TCC_ATOMIC / TCC_EA0_ATOMIC per launch), profiles/r01_pmc_calibration.txt (the probe ceiling)"}
        elif achieved and form == "fused" and k == 64:
            # every processed triplet issues 10.04 64-byte fp32 atomic requests (TCC_ATOMIC counters, profiles/r02_sgd_pmc.csv:
            # all forwarded to the memory side); the chip retires ~20 G such requests/s in tools/atomic_probe.hip
            req = (n_draws - skipped) / max(launches, 1) * 10.04 / avg_launch_s / 1e9
            out["roofline"]["limiter"] = {"what": "memory-side fp32 atomic requests (64 B granules)", "requests_per_triplet": 10.04,
                                          "achieved_G_per_s": req, "probe_ceiling_G_per_s": 20.0, "frac_of_probe_ceiling": req / 20.0,
                                          "evidence": "profiles/r02_sgd_pmc.csv, profiles/r01_pmc_calibration.txt"}
        out["train_stats"] = {"correct_frac": correct / max(n_draws - skipped, 1.0), "skipped_frac": skip_frac}

    # ---- scoring leg: batched rank() over users with fused top-10 --------------------------------------------
    full_items = None
    if not args.no_rank and distributed and args.sharded_items:
        Vt, Bt = sharded.table.gather_full()  # collective: every rank assembles the sharded item table
        full_items = (Vt.cpu().numpy(), Bt.cpu().numpy())
    if not args.no_rank and rank == 0:
        if full_items is not None:
            U2, (V2, B2) = trainer.get_user_factors(), full_items
        else:
            U2, V2, B2 = trainer.get_factors()
        sc = _lib.Scorer(U2, V2, B2, None, device=local_rank)
        n_rank = n_users if args.rank_users <= 0 else min(args.rank_users, n_users)
        # The evaluation protocol's ranking (cornac/eval_methods/base_method.py:176-220): every user's top-10 with the
        # TRAINING POSITIVES EXCLUDED, results copied back to the host.  The exclusion lists (the training CSR) are
        # registered once and stay on the device, as they do across the epochs / models evaluated on one split.
        sc.set_exclusions(indptr.astype(np.int64), indices)
        sc.rank_topk_resident((0, n_rank), 10, fetch="items", pinned=True)  # warm-up at full size: the device workspaces and the page-locked result buffer are allocated here, not in the timed call
        # (median of five calls: evaluation ranks chunk after chunk, and single calls spread by +-5 % with the clock ramp)
        walls, devs = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            items, _, ms_dev = sc.rank_topk_resident((0, n_rank), 10, fetch="items", timed=True, pinned=True)
            walls.append(1e3 * (time.perf_counter() - t0))
            devs.append(ms_dev)
        ms, ms_dev = float(np.median(walls)), float(np.median(devs))
        assert items.shape == (n_rank, 10)
        # the same ranking with the lists handed over per call (H2D of the 80 MB CSR included) and without exclusions
        t0 = time.perf_counter()
        sc.rank_topk(np.arange(n_rank, dtype=np.int32), 10, exclude=(indptr[:n_rank + 1].astype(np.int64), indices[:indptr[n_rank]]))
        ms_percall = 1e3 * (time.perf_counter() - t0)
        sc.rank_topk_device_ms(0, n_rank, 10, 1)
        ms_plain = sc.rank_topk_device_ms(0, n_rank, 10, 1)
        # full ranking (rank(k=-1)): materialised score tile + per-row sort, 10 000 users
        n_full = min(args.rank_full_users, n_users)
        ms_full = None
        if n_full > 0:
            sc.rank_topk_device_ms(0, min(n_full, 512), n_items, 1)
            ms_full = sc.rank_topk_device_ms(0, n_full, n_items, 1)
        pairs = float(n_rank) * n_items
        out["rank"] = {"metric": "rank_items_scored_per_sec", "value": pairs / (ms / 1e3), "unit": "items/s",
                       "users": n_rank, "items": n_items, "topk": 10, "ms": ms,
                       "what": "top-10 of every user with the user's training positives excluded (resident lists), "
                               "the ranked item ids copied to the host (what the @k metrics of ranking_eval read): wall time of the call, median of 5",
                       "ms_calls": walls,
                       "device_ms": ms_dev, "ms_lists_passed_per_call": ms_percall,
                       "ms_no_exclusions_device_only": ms_plain,
                       "roofline": {"bound": "mfma", "achieved": 2.0 * k * pairs / (ms / 1e3) / 1e12,
                                    "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                    "frac": 2.0 * k * pairs / (ms / 1e3) / 1e12 / FP32_MFMA_PEAK_TF,
                                    "frac_device_only": 2.0 * k * pairs / (ms_dev / 1e3) / 1e12 / FP32_MFMA_PEAK_TF,
                                    "frac_no_exclusions_device_only": 2.0 * k * pairs / (ms_plain / 1e3) / 1e12 / FP32_MFMA_PEAK_TF}}
        if ms_full is not None:
            out["rank"]["full_ranking"] = {"topk": -1, "users": n_full, "ms": ms_full,
                                           "value": float(n_full) * n_items / (ms_full / 1e3), "unit": "items/s",
                                           "note": "rank(k=-1): score tile materialised, every row fully sorted"}
        sc.close()
        if world == 1 and args.cpu_baseline_seconds > 0:
            try:
                out["rank"]["cpu_baseline"] = cpu_rank_baseline(U2, V2, B2, 10)
            except Exception as e:
                print("[bench] rank cpu_baseline failed: %r" % (e,), file=sys.stderr)
                out["rank"]["cpu_baseline"] = None
    trainer.close()

    # ---- extra single-GPU legs (rank 0, N = 1 only) -------------------------------------------------------------
    if rank == 0 and world == 1 and not distributed and not args.no_legs and args.legs:
        out["legs"] = {}
        for name in [x for x in args.legs.split(",") if x]:
            t_leg = time.time()
            try:
                out["legs"][name] = LEGS[name](args, _lib)
                out["legs"][name]["leg_wall_s"] = time.time() - t_leg
            except Exception as e:  # a leg must not take the headline line down with it
                print("[bench] leg %s failed: %r" % (name, e), file=sys.stderr)
                out["legs"][name] = {"error": repr(e)}

    # ---- CPU baseline leg (rank 0, N = 1 only) ------------------------------------------------------------------
    if rank == 0 and world == 1 and args.cpu_baseline_seconds > 0:
        try:
            if scale:  # the reference kernel on the first 1 M users of the slice (its U slice is 0.5 GB instead of 6.4 GB)
                sub = 1_000_000
                out["cpu_baseline"] = cpu_baseline(indptr[:sub + 1], indices[:indptr[sub]], n_items, k, args.lr, args.reg,
                                                   min(args.cpu_baseline_seconds, 8.0))
                out["cpu_baseline"]["sample"] = ("first %d users of the slice; " % sub +
                                                 out["cpu_baseline"]["sample"].replace("ML-20M-shaped", "scale-config"))
            else:
                out["cpu_baseline"] = cpu_baseline(indptr, indices, n_items, k, args.lr, args.reg, args.cpu_baseline_seconds)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        except Exception as e:  # the measured GPU line must not be lost to a host-side baseline problem
            print("[bench] cpu_baseline failed: %r" % (e,), file=sys.stderr)
            out["cpu_baseline"] = None
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    emit_json_line(out if rank == 0 else None)


def emit_json_line(out):
    """ONE JSON line on stdout, and nothing else: native libraries (RCCL prints a version banner through C stdio when a
    communicator is created) hold text in their own stdout buffer that would otherwise be flushed AFTER the line at process
    exit.  Their pending text goes to stderr before the line, and stdout is pointed at stderr afterwards."""
    import ctypes

    sys.stdout.flush()
    libc = ctypes.CDLL(None)
    keep = os.dup(1)
    os.dup2(2, 1)
    libc.fflush(None)
    os.dup2(keep, 1)
    os.close(keep)
    if out is not None:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()
    os.dup2(2, 1)


if __name__ == "__main__":
    main()
