#!/usr/bin/env python3
"""End-to-end latency of a hogwild BPR fit at ML-20M shape: where the time goes outside the SGD kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import _lib, synth
n_users, n_items, nnz, a, seed = synth.CONFIGS["ml20m"]
users, items = synth.zipf_interactions(n_users, n_items, nnz, a, seed)
indptr, indices = synth.csr_from_sorted(users, items, n_users)
k = 64
rs = np.random.RandomState(0)
U = ((rs.uniform(0, 1, (n_users, k)) - .5) / k).astype(np.float32); V = ((rs.uniform(0, 1, (n_items, k)) - .5) / k).astype(np.float32)
t0 = time.perf_counter(); tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k); t1 = time.perf_counter()
tr.set_factors(U, V, np.zeros(n_items, np.float32)); tr.seed_hogwild(1); t2 = time.perf_counter()
tr.fit_epochs(1, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD); t3 = time.perf_counter()
tr.fit_epochs(19, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD); t4 = time.perf_counter()
out = tr.get_factors(); t5 = time.perf_counter()
print("create %.3f s | set_factors %.3f | first epoch (builds ownership) %.3f | 19 epochs %.3f | get_factors %.3f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
