#!/usr/bin/env python3
"""CPU simulation behind the LDS-bin sampling contract: BPR hogwild on the host (OpenMP, the reference's
arithmetic) at the ML-20M shape with the negative drawn uniformly over ALL items (bins = 1) or uniformly inside the
positive's bin, bins re-dealt every epoch by popularity-rank groups.  Prints the pairwise loss / accuracy on a fixed
probe sample after 5 / 10 / 20 / 30 epochs.      python tools/sim_binned_negatives.py 1,8,256,1280"""
import subprocess
import sys, os, time, ctypes as C, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench
n_users, n_items, indptr, indices = bench.load_dataset("ml20m", 0, os.environ.get("TMPDIR", "/tmp"))
nnz = len(indices); k = 64
user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
deg = np.bincount(indices, minlength=n_items)
rank_item = np.argsort(-deg, kind="stable").astype(np.int32)
item_rank = np.empty(n_items, np.int32); item_rank[rank_item] = np.arange(n_items, dtype=np.int32)
rs = np.random.RandomState(1); pick = rs.randint(nnz, size=400000)
pu, pi, pj = user_ids[pick], indices[pick], rs.randint(n_items, size=400000)
def probe(U, V, B):
    x = B[pi] - B[pj] + np.einsum("nk,nk->n", U[pu], V[pi] - V[pj]); return float(np.mean(np.log1p(np.exp(-x)))), float(np.mean(x > 0))
so = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libsim_binned.so")
subprocess.check_call(["gcc", "-O3", "-ffast-math", "-fopenmp", "-shared", "-fPIC", os.path.join(HERE, "sim_binned_negatives.c"), "-o", so])
L = C.CDLL(so)
i32 = np.ctypeslib.ndpointer(np.int32, flags="C"); f32 = np.ctypeslib.ndpointer(np.float32, flags="C")
L.sim_epochs.argtypes = [i32, i32, i32, i32, i32, C.c_int64, C.c_int, f32, f32, f32, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint64, C.c_int]
for nbins in [int(x) for x in sys.argv[1].split(",")]:
    U, V, B = bench.init_factors(n_users, n_items, k, 100)
    done = 0; out = []
    t0 = time.time()
    for e in (5, 10, 20, 30):
        L.sim_epochs(indptr, indices, user_ids, rank_item, item_rank, nnz, n_items, U, V, B, k, 0.05, 0.01, nbins, e - done, 7, done); done = e
        out.append("e%d (%.4f, %.4f)" % ((e,) + probe(U, V, B)))
    print("bins %5d: %s  [%.0f s]" % (nbins, "  ".join(out), time.time() - t0), flush=True)
