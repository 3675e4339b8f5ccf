#!/bin/bash
# round 4: smoke() and the LDS-bin tests on the library as committed
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 60 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
timeout 60 python -m pytest tests/test_bpr_gpu.py -q -m gpu -k "ldsbin" --timeout 50 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror" | tail -3
