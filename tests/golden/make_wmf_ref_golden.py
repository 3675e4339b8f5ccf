#!/usr/bin/env python3
"""Generates tests/golden/wmf_ref.npz by running the reference's OWN WMF (cornac/models/wmf/recom_wmf.py + wmf.py,
unmodified; build container only).  TensorFlow is absent from the image, so the two files run over oracle/tf1_shim — a
stand-in for the tensorflow.compat.v1 symbols they use with torch underneath (forward values and autograd of the loss the
reference builds; the gather-gradient / clip / TF1 Adam rules are restated there, see its README).  With a real
TensorFlow importable the script uses that instead and says so in the fixture (`backend`).

Stored: the interactions in insertion order, the hyper-parameters, and what the reference learned from its own xavier
initialisation (seed) and its own item_iter shuffling (data set seed): U, V, score() of three users.

    python tests/golden/make_wmf_ref_golden.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

from oracle import ref_loader, ref_wmf  # noqa: E402


def main():
    RefWMF = ref_wmf.load_wmf()
    ns = ref_loader.load()
    rs = np.random.RandomState(21)
    nu, ni, nnz = 120, 64, 1400
    keys = rs.permutation(nu * ni)[:nnz]
    u, i, r = keys // ni, keys % ni, rs.randint(1, 6, nnz).astype(np.float64)
    ds = ns.Dataset.from_uir([(int(a), int(b), float(c)) for a, b, c in zip(u, i, r)], seed=123)
    kw = dict(k=10, max_iter=5, batch_size=24, learning_rate=0.005, lambda_u=0.02, lambda_v=0.03, a=1.0, b=0.02, seed=9)
    m = RefWMF(verbose=False, **kw).fit(ds)
    users = np.array([0, nu // 2, ds.num_users - 1], np.int64)
    np.savez_compressed(os.path.join(HERE, "wmf_ref.npz"), users=u, items=i, ratings=r, U=m.U, V=m.V, score_users=users,
                        scores=np.stack([m.score(int(x)) for x in users]),
                        backend=np.array("tf1_shim(torch)" if ref_wmf.uses_shim() else "tensorflow"),
                        **{n: np.float64(v) for n, v in kw.items()})
    print("wrote wmf_ref", m.U.dtype, m.U.shape, m.V.shape, "backend:", "shim" if ref_wmf.uses_shim() else "tensorflow")


if __name__ == "__main__":
    main()
