import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cornac_amd import _lib, synth
n_users, n_items, nnz, a, seed = synth.CONFIGS["ml20m"]
users, items = synth.zipf_interactions(n_users, n_items, nnz, a, seed)
indptr, indices = synth.csr_from_sorted(users, items, n_users)
k = 64
tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
rs = np.random.RandomState(0)
tr.set_factors(((rs.uniform(0,1,(n_users,k))-.5)/k).astype(np.float32), ((rs.uniform(0,1,(n_items,k))-.5)/k).astype(np.float32), np.zeros(n_items, np.float32))
tr.seed_mt19937(123, 456)
for e in range(3):
    t0 = time.perf_counter()
    c, s = tr.fit_epochs(1, 0.05, 0.01, True, _lib.NEG_UNIFORM, _lib.MODE_DETERMINISTIC)
    print("epoch", e, "wall %.3f s" % (time.perf_counter() - t0), tr.last_timing(), c, s)
