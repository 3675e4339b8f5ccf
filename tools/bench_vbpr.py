#!/usr/bin/env python3
"""VBPR throughput probe at the Tradesy shape (BASELINE.json configs[3]): 19 243 users x 165 906 items,
394 421 feedback, 4096-d visual features, k = k2 = 64, batch 100.  Times the device part of one epoch
(forward / B x B objective / scatter / feature-GEMM + Adam / dense Adam) on pre-sampled batches."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import _lib

nu, ni, nnz, nf, k, k2, B = 19243, 165906, 394421, 4096, 64, 64, 100
scale = float(os.environ.get("VBPR_SCALE", "1.0"))
ni = int(ni * scale)
rs = np.random.RandomState(44)
F = rs.uniform(0, 1, (ni, nf)).astype(np.float32)
u = rs.randint(0, nu, nnz).astype(np.int32); i = rs.randint(0, ni, nnz).astype(np.int32); j = rs.randint(0, ni, nnz).astype(np.int32)
tr = _lib.VbprTrainer(F, nu, ni, k, k2)
lim = np.sqrt(3.0) * np.sqrt(2.0 / (nu + k))
tr.set_params(Bi=np.zeros(ni, np.float32), Gu=rs.uniform(-lim, lim, (nu, k)), Gi=rs.uniform(-lim, lim, (ni, k)),
              Tu=rs.uniform(-lim, lim, (nu, k2)), E=rs.uniform(-0.03, 0.03, (nf, k2)), Bp=rs.uniform(-0.03, 0.03, nf))
tr.fit_batches(u[:2000], i[:2000], j[:2000], B, 0.005, 0.01, 0.01, 0.0)
t0 = time.perf_counter()
nll = tr.fit_batches(u, i, j, B, 0.005, 0.01, 0.01, 0.0)
dt = time.perf_counter() - t0
steps = (nnz + B - 1) // B
n_par = ni + nu * k + ni * k + nu * k2 + nf * k2 + nf
print(json.dumps({"shape": [nu, ni, nnz, nf], "k": k, "k2": k2, "batch": B, "epoch_s": dt, "steps": steps,
                  "us_per_step": 1e6 * dt / steps, "triplets_per_s": nnz / dt,
                  "adam_bytes_per_step": 28 * n_par, "adam_sweep_GBps": 28 * n_par * steps / dt / 1e9,
                  "mean_nll_per_pair": nll / (steps * B * B)}))
