#!/bin/bash
# round 4: the LDS-bin deal (strata-permuted groups + levelled hot runs): tests, then a sweep of the deal parameters
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bpr_gpu.py -x -q -m gpu -k "ldsbin" > gpurun_out/ldsbin_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/ldsbin_tests.log
tail -15 gpurun_out/ldsbin_tests.log
CORNAC_HIP_PROFILE=1 timeout 1800 python tools/exp_strata.py --cpu-threads 0 --epochs ${EPOCHS:-10} --report ${REPORT:-5,10} --arms "${ARMS:-ldsbin}" > gpurun_out/exp_ldsbin.log 2>&1
echo "exp rc=$?" >> gpurun_out/exp_ldsbin.log
cat gpurun_out/exp_ldsbin.log
