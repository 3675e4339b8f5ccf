#!/usr/bin/env python3
"""A/B of the fused top-k kernel's exclusion bitmap against brute force on random data: prints what differs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornac_amd import _lib

def run(nu, ni, k, deg, seed):
    rs = np.random.RandomState(seed)
    U = rs.normal(0, 0.3, (nu, k)).astype(np.float32); V = rs.normal(0, 0.3, (ni, k)).astype(np.float32)
    B = rs.normal(0, 0.3, ni).astype(np.float32)
    lists = [np.sort(rs.choice(ni, size=min(ni - 11, rs.randint(0, 2 * deg)), replace=False)).astype(np.int32) for _ in range(nu)]
    ptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    idx = np.concatenate(lists) if ptr[-1] else np.empty(0, np.int32)
    sc = _lib.Scorer(U, V, B, None)
    users = np.arange(nu, dtype=np.int32)
    items, _ = sc.rank_topk(users, 10, exclude=(ptr, idx))
    S = sc.score_block(users)
    bad = 0
    for r in range(nu):
        s = S[r].copy(); s[lists[r]] = -np.inf
        want = np.lexsort((np.arange(ni), s))[::-1][:10]
        if not np.array_equal(items[r], want):
            bad += 1
            if bad <= 5:
                miss = [int(x) for x in want if x not in items[r]]
                extra = [int(x) for x in items[r] if x not in want]
                owners = {m: [q for q in range(max(0, r - r % 32), min(nu, r - r % 32 + 32)) if m in lists[q]] for m in miss}
                print("  row", r, "missing", miss, "extra", extra, "extra excluded?", [int(e in lists[r]) for e in extra],
                      "missing items are excluded for rows of the same 32-row tile:", owners)
    sc.close()
    print("nu %d ni %d k %d deg %d: %d / %d rows differ" % (nu, ni, k, deg, bad, nu))

for args in [(300, 500, 16, 20, 0), (300, 5000, 64, 100, 1), (1000, 26744, 64, 144, 2), (64, 70, 8, 10, 3)]:
    run(*args)
