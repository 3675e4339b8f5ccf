#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
CORNAC_HIP_PROFILE=1 timeout 900 python tools/strata_census.py > gpurun_out/census.log 2>&1
echo "census rc=$?" >> gpurun_out/census.log
CORNAC_HIP_PROFILE=1 timeout 1500 python tools/exp_strata.py --cpu-threads 0 --arms "${ARMS:-atomic,strata}" > gpurun_out/exp_strata2.log 2>&1
echo "exp rc=$?" >> gpurun_out/exp_strata2.log
cat gpurun_out/census.log gpurun_out/exp_strata2.log
