#!/bin/bash
# round 4: the MF minibatch path with dropout on the device (against the real reference's goldens and the oracle)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_mf_minibatch.py -q -m gpu --timeout 100 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert |Timeout|^E " | cut -c1-300 | tail -12
