#!/usr/bin/env python3
"""dist.fit_bpr_sharded / fit_mf_sharded on ONE rank through a real RCCL group (device path: bound item table, driver
stream, all-gathers of size one) next to the plain model.fit on the same data: both must learn."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cornac_amd as ca  # noqa: E402
from cornac_amd import synth  # noqa: E402
from cornac_amd.dist import fit_bpr_sharded, fit_mf_sharded  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
users, items = synth.zipf_interactions(4000, 1500, 300_000, 0.7, 3)
rs = np.random.RandomState(0)
P, Q = rs.normal(0, 1, (4000, 4)), rs.normal(0, 1, (1500, 4))
val = np.clip(np.rint(3.0 + 0.6 * np.einsum("nk,nk->n", P[users], Q[items]) + rs.normal(0, 0.3, len(users))), 1, 5)
ds = ca.Dataset.from_uir(list(zip(users.tolist(), items.tolist(), val.tolist())), seed=1)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
try:
    kw = dict(k=64, max_iter=6, learning_rate=0.05, lambda_reg=0.01, seed=5, mode="hogwild")
    a = fit_bpr_sharded(ca.BPR(**kw), ds, device=dev, sync_per_epoch=8)
    b = ca.BPR(**kw).fit(ds)
    fa = a.fit_stats[0][0] / max(6 * ds.matrix.nnz - a.fit_stats[0][1], 1)
    fb = b.fit_stats[0][0] / max(6 * ds.matrix.nnz - b.fit_stats[0][1], 1)
    print("BPR correct fraction: sharded fit %.4f, plain fit %.4f; |U| %.3f vs %.3f" % (fa, fb, np.abs(a.u_factors).mean(), np.abs(b.u_factors).mean()))
    assert np.isfinite(a.u_factors).all() and abs(fa - fb) < 0.03 and fa > 0.6
    kw = dict(k=64, max_iter=8, learning_rate=0.01, lambda_reg=0.02, seed=5, mode="hogwild")
    m = fit_mf_sharded(ca.MF(**kw), ds, device=dev, parts_per_epoch=8)
    p = ca.MF(**kw).fit(ds)
    print("MF loss: sharded fit %s, plain fit %s" % (np.round(m.loss_history[[0, -1]], 1), np.round(np.asarray(p.loss_history)[[0, -1]], 1)))
    assert m.loss_history[-1] < m.loss_history[0] and abs(m.loss_history[-1] - p.loss_history[-1]) < 0.15 * p.loss_history[-1]
    print("ok")
finally:
    dist.destroy_process_group()
