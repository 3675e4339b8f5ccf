#!/usr/bin/env python3
"""CPU simulation behind the LDS-bin sampling contract: BPR hogwild on the host (OpenMP, the reference's
arithmetic) at the ML-20M shape with the negative drawn uniformly over ALL items ("global") or uniformly inside the
positive's bin, the bins dealt every epoch by the restated deal of the HIP path (oracle.ldsbin_deal_key): `bins:G` =
`bins` bins, strata of G groups (G = 1 is round 3's deal: static rank groups, two items of a group never meet).
Two probes on fixed samples after 5 / 10 / 20 / 30 epochs, as (loss, accuracy): j uniform over all items, and j drawn
among the 2 x 128 popularity-rank neighbours of i (the pairs the static groups excluded).
      python tools/sim_binned_negatives.py global,256:1,256:16"""
import subprocess
import sys, os, time, ctypes as C, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench
from oracle import oracle as orc
n_users, n_items, indptr, indices = bench.load_dataset("ml20m", 0, os.environ.get("TMPDIR", "/tmp"))
nnz = len(indices); k = 64
user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
deg = np.bincount(indices, minlength=n_items)
rank_item = np.argsort(-deg, kind="stable").astype(np.int32)
item_rank = np.empty(n_items, np.int32); item_rank[rank_item] = np.arange(n_items, dtype=np.int32)
cptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
rs = np.random.RandomState(1); N = 400000
pick = rs.randint(nnz, size=N)
pu, pi, pj = user_ids[pick], indices[pick], rs.randint(n_items, size=N)
delta = rs.randint(1, 129, size=N) * rs.choice([-1, 1], size=N)
pj_near = rank_item[np.clip(item_rank[pi] + delta, 0, n_items - 1)]
def is_pos(u, j):
    key = u.astype(np.int64) * n_items + j
    allk = user_ids.astype(np.int64) * n_items + indices
    return np.isin(key, allk)
ok_near = ~is_pos(pu, pj_near) & (pj_near != pi)
def probe(U, V, B, j, ok=None):
    x = B[pi] - B[j] + np.einsum("nk,nk->n", U[pu], V[pi] - V[j])
    if ok is not None: x = x[ok]
    return float(np.mean(np.log1p(np.exp(-x)))), float(np.mean(x > 0))
so = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libsim_binned.so")
subprocess.check_call(["gcc", "-O3", "-ffast-math", "-fopenmp", "-shared", "-fPIC", os.path.join(HERE, "sim_binned_negatives.c"), "-o", so])
L = C.CDLL(so)
i32 = np.ctypeslib.ndpointer(np.int32, flags="C"); f32 = np.ctypeslib.ndpointer(np.float32, flags="C")
L.sim_epoch.argtypes = [i32, i32, i32, i32, i32, i32, C.c_int64, C.c_int, f32, f32, f32, C.c_int, C.c_float, C.c_float, C.c_int, C.c_uint64, C.c_int]
for spec in sys.argv[1].split(","):
    nbins, groups = (1, 1) if spec == "global" else [int(x) for x in spec.split(":")]
    U, V, B = bench.init_factors(n_users, n_items, k, 100)
    out = []
    t0 = time.time()
    n_strata = orc.ldsbin_n_strata(n_items, nbins, groups)
    for e in range(30):
        if nbins > 1:
            key = int(orc.lib().oracle_ldsbin_key(7, e))
            bin_of = orc.ldsbin_deal_key(key, nbins, n_items, 0, n_strata, 0, rank_item, cptr, 0)[0]
            order = np.argsort(bin_of, kind="stable").astype(np.int32)
            bptr = np.concatenate([[0], np.cumsum(np.bincount(bin_of, minlength=nbins))]).astype(np.int32)
        else:
            bin_of = np.zeros(n_items, np.int32); order = np.arange(n_items, dtype=np.int32); bptr = np.array([0, n_items], np.int32)
        L.sim_epoch(indptr, indices, user_ids, bin_of, bptr, order, nnz, n_items, U, V, B, k, 0.05, 0.01, nbins, 7, e)
        if e + 1 in (5, 10, 20, 30):
            out.append("e%d all (%.4f, %.4f) near (%.4f, %.4f)" % ((e + 1,) + probe(U, V, B, pj) + probe(U, V, B, pj_near, ok_near)))
    print("%-8s: %s  [%.0f s]" % (spec, "  ".join(out), time.time() - t0), flush=True)
