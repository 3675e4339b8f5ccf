import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cornac_amd import _lib
rs = np.random.RandomState(0)
nu, ni, k = 4096, 2_000_000, 128
U = rs.normal(0, .3, (nu, k)).astype(np.float32); V = rs.normal(0, .3, (ni, k)).astype(np.float32)
b = rs.normal(0, .1, ni).astype(np.float32)
t0 = time.perf_counter(); sc = _lib.Scorer(U, V, b, None); t1 = time.perf_counter()
users = np.arange(nu, dtype=np.int32)
items, scores = sc.rank_topk(users, 10); t2 = time.perf_counter()
ms = sc.rank_topk_device_ms(0, nu, 10, 1)
for u in (0, 17, 4095):
    s = (V.astype(np.float64) @ U[u].astype(np.float64)) + b
    top = np.argsort(-s)[:10]
    assert set(top) == set(items[u].tolist()), (u, top, items[u])
print("scorer set %.2f s, rank %.3f s (device %.1f ms), %.1f TFLOP/s" % (t1 - t0, t2 - t1, ms, 2.0 * nu * ni * k / ms / 1e9))
