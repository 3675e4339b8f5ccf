#!/bin/bash
# first GPU session of the strata form: sampler exactness tests, then the A/B sweep (profile build)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bpr_gpu.py -x -q -m gpu -k "strata or owned or atomic_updates" > gpurun_out/strata_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/strata_tests.log
CORNAC_HIP_PROFILE=1 timeout 1500 python tools/exp_strata.py --arms "${ARMS:-atomic,strata}" > gpurun_out/exp_strata.log 2>&1
echo "exp rc=$?" >> gpurun_out/exp_strata.log
tail -5 gpurun_out/strata_tests.log
cat gpurun_out/exp_strata.log
