#!/usr/bin/env python3
"""What each piece of the WMF user step costs (profile build: CORNAC_HIP_PROFILE=1, CORNAC_HIP_WMF_ABLATE bits — 1 the P = U Vb^T
product, 2 the dV = G^T U product, 4 the dU = G Vb product, 8 the Adam epilogue, 16 the non-zeros' fix-up; results of an ablated
run are garbage, only its time counts).   CORNAC_HIP_PROFILE=1 python tools/wmf_ablate.py [--users 480189 --items 17770]"""
import argparse, json, os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=480189)
ap.add_argument("--items", type=int, default=17770)
ap.add_argument("--density", type=float, default=0.0118)
ap.add_argument("--k", type=int, default=128)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--masks", default="0,1,2,4,8,16,3,7,15,31")
ap.add_argument("--zeros", action="store_true", help="zero factors: the same instruction stream at a fraction of the switching power (DVFS check)")
ap.add_argument("--variants", default="", help="CORNAC_HIP_WMF_VARIANT values to time at mask 0 (wmf.hip: VAR), e.g. 0,1,2,4,8:2000,9:1500")
args = ap.parse_args()
rs = np.random.RandomState(0)
n_cols = 128 * (args.steps + 3)           # only the columns the timed batches touch carry ratings
nnz = int(args.users * n_cols * args.density)
keys = np.unique(rs.randint(0, args.users * n_cols, size=int(nnz * 1.02), dtype=np.int64))
users, items = keys // n_cols, keys % n_cols
val = rs.randint(1, 6, len(users)).astype(np.float32)
R = sp.csc_matrix((val, (users, items)), shape=(args.users, args.items))
k = args.k
tr = _lib.WmfTrainer(R, k)
lim = np.sqrt(6.0 / (args.users + k))
U0 = rs.uniform(-lim, lim, (args.users, k)).astype(np.float32)
V0 = rs.uniform(-lim, lim, (args.items, k)).astype(np.float32)
if args.zeros:
    U0[:] = 0
    V0[:] = 0
batches = [np.arange(s, s + 128, dtype=np.int32) for s in range(0, n_cols, 128)]
out = {}
for m in [int(x) for x in args.masks.split(",")]:
    os.environ["CORNAC_HIP_WMF_ABLATE"] = str(m)
    tr.set_factors(U0, V0)
    tr.fit_batches(batches[:3], 0.01, 0.01, 1.0, 0.01, 0.001)
    tr.kernel_timing(True)
    tr.fit_batches(batches[3:], 0.01, 0.01, 1.0, 0.01, 0.001)
    out[m] = tr.last_device_ms() / len(batches[3:])
    print("ablate %2d: %.3f ms per step" % (m, out[m]), flush=True)
var = {}
os.environ["CORNAC_HIP_WMF_ABLATE"] = "0"
for spec in [x for x in args.variants.split(",") if x]:
    v, ticks, mask = (spec.split(":") + ["", ""])[:3]     # variant[:stagger ticks[:ablation mask]]
    os.environ["CORNAC_HIP_WMF_VARIANT"] = v
    os.environ["CORNAC_HIP_WMF_ABLATE"] = mask or "0"
    if ticks:
        os.environ["CORNAC_HIP_WMF_STAGGER"] = ticks
    tr.set_factors(U0, V0)
    tr.fit_batches(batches[:3], 0.01, 0.01, 1.0, 0.01, 0.001)
    tr.kernel_timing(True)
    best = 1e9
    for rep in range(2):
        tr.fit_batches(batches[3:], 0.01, 0.01, 1.0, 0.01, 0.001)
        best = min(best, tr.last_device_ms() / len(batches[3:]))
    var[spec] = best
    print("variant %-8s: %.3f ms per step  (frac %.3f)" % (spec, best, 6.0 * args.users * 128 * k / (best / 1e3) / 157.3e12), flush=True)
flops = 6.0 * args.users * 128 * k
print(json.dumps({"ms_per_step": out, "variants": var, "mfma_frac_full": flops / (out.get(0, 1) / 1e3) / 157.3e12}))
