"""The real compiled reference kernel (oracle/_ref) must load and run over the stand-in Python modules of
oracle/ref_stubs when the reference tree is absent (the GPU box): that is what bench.py's
`cpu_baseline.kind = "reference"` times.  Runs in a subprocess so that no `cornac` package of this process interferes."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from oracle import ref_loader

SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import ref_loader
assert not ref_loader.available()
RNGVector, BPR = ref_loader.load_kernel_only()
rs = np.random.RandomState(0)
nu, ni, k = 500, 200, 8
deg = rs.randint(3, 20, nu)
indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
indices = np.concatenate([np.sort(rs.choice(ni, d, replace=False)) for d in deg]).astype(np.int32)
user_ids = np.repeat(np.arange(nu), deg).astype(np.int32)
U = ((rs.uniform(0, 1, (nu, k)) - .5) / k).astype(np.float32)
V = ((rs.uniform(0, 1, (ni, k)) - .5) / k).astype(np.float32)
B = np.zeros(ni, np.float32)
V0 = V.copy()
c, s = BPR(k=k, learning_rate=0.05)._fit_sgd(RNGVector(1, len(indices) - 1, 1), RNGVector(1, ni - 1, 2), 1, user_ids, indices,
                                             np.arange(ni, dtype=np.int32), indptr, U, V, B)
assert 0 < c <= len(indices) and 0 <= s < len(indices) and np.abs(V - V0).max() > 0
print("OK", c, s)
'''


@pytest.mark.skipif(not ref_loader.kernel_available(), reason="oracle/_ref not built")
def test_compiled_reference_kernel_runs_over_the_stubs():
    env = dict(os.environ, CORNAC_REFERENCE="/nonexistent-reference")
    out = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]
