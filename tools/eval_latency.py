#!/usr/bin/env python3
"""Wall time of a full ranking evaluation (Recall@10, NDCG@10) at ML-20M shape: 138 493 users, 26 744 items, the
training positives excluded per user — the reference's "Test (s)" column (one Python rank() call per user there);
then the full-list metrics (AUC, MAP, MRR) over the same users: the device counts where each test positive stands
among the user's candidates (cornac_hip_rank_positions), no ranking is produced."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import BPR, Dataset, eval as ev, metrics as mm, synth

n_users, n_items, nnz, a, seed = synth.CONFIGS["ml20m"]
users, items = synth.zipf_interactions(n_users, n_items, nnz, a, seed)
rs = np.random.RandomState(0)
test_sel = rs.rand(len(users)) < 0.05
tr = Dataset.from_arrays(users[~test_sel], items[~test_sel], np.ones((~test_sel).sum()), n_users, n_items)
te = Dataset.from_arrays(users[test_sel], items[test_sel], np.ones(test_sel.sum()), n_users, n_items)
t0 = time.perf_counter()
m = BPR(k=64, max_iter=10, learning_rate=0.05, lambda_reg=0.01, verbose=False).fit(tr)
t1 = time.perf_counter()
(recall, ndcg), _ = ev.ranking_eval(m, [mm.Recall(k=10), mm.NDCG(k=10)], tr, te)
t2 = time.perf_counter()
print("fit (10 hogwild epochs, incl. setup) %.2f s | ranking_eval over %d test users %.2f s | Recall@10 %.4f NDCG@10 %.4f"
      % (t1 - t0, len(set(users[test_sel].tolist())), t2 - t1, recall, ndcg))
t3 = time.perf_counter()
(auc, mapv, mrr), per_user = ev.ranking_eval(m, [mm.AUC(), mm.MAP(), mm.MRR()], tr, te)
t4 = time.perf_counter()
print("full-list ranking_eval (AUC, MAP, MRR; device position counts) over %d test users %.2f s | AUC %.4f MAP %.4f MRR %.4f"
      % (len(per_user[0]), t4 - t3, auc, mapv, mrr))
