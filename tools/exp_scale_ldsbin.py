#!/usr/bin/env python3
"""configs[4] one-GPU slice: the XCD-strata form against the LDS-bin form with PASSING bins (the item table passes
through the LDS once per epoch).  python tools/exp_scale_ldsbin.py [variant ...]; a variant is a comma list of
waves=<4|8|16>, kb=<LDS KiB per bin>, unr=<triplets in flight per wave>, abl=<ablation bits>, epochs=<n> — the last
three need the profile build (CORNAC_HIP_PROFILE=1).  "strata" = the XCD-strata form."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cornac_amd import _lib

k = bench.SCALE["k"]
nu, ni, indptr, indices = bench.scale_slice(0)
nnz = len(indices)
U, V, B = bench.scale_factors(nu, ni, k, 0)
b_full, _ = bench.algorithmic_bytes_per_triplet(k, bench.SCALE["degree"])
for spec in (sys.argv[1:] or ["strata", "waves=8,kb=78"]):
    cfg = dict(kv.split("=") for kv in spec.split(",")) if spec != "strata" else {}
    tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
    flags = 0
    if spec == "strata":
        tr.ldsbin_pass_config(False)
    else:
        tr.ldsbin_pass_config(True, int(cfg.get("waves", 8)), int(cfg.get("kb", 78)))
        os.environ["CORNAC_HIP_LDSBIN_UNR"] = cfg.get("unr", "0")
        flags = int(cfg.get("abl", 0)) << 8
    tr.set_factors(U, V, B)
    tr.seed_hogwild(7)
    t0 = time.time()
    tr.fit_epochs(1, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags)
    t_first = time.time() - t0
    st = tr.ldsbin_stats() if spec != "strata" else {}
    n_ep = int(cfg.get("epochs", 3))
    tr.kernel_timing(True)
    t0 = time.perf_counter()
    c, sk = tr.fit_epochs(n_ep, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags)
    dt = (time.perf_counter() - t0) / n_ep
    kms, launches = tr.kernel_timing(False)
    print(json.dumps({"variant": spec, "ms_per_epoch": round(1e3 * dt, 3), "kernel_ms_per_epoch": round(kms / n_ep, 3),
                      "launches": launches, "frac": round(nnz * b_full / dt / 8e12, 4), "correct": c / max(n_ep * nnz - sk, 1),
                      "skipped": sk / (float(n_ep) * nnz), "first_epoch_s": round(t_first, 2), "stats": st}), flush=True)
    tr.close()
