#!/bin/bash
# round 4, call 6: strata kernel at the configs[4] slice with streaming (nt) stores of the user / item rows, and without
# the bias traffic (profile build ablation bits: 64 U stores nt, 128 V stores nt, 8 no bias)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for abl in 0 64 128 192 8 0; do
  CORNAC_HIP_PROFILE=1 timeout 600 python bench.py --config scale --steps 3 --warmup 1 --cpu-baseline-seconds 0 --flags $((abl << 8)) > gpurun_out/r04_scale_nt$abl.json.log 2> gpurun_out/r04_scale_nt$abl.err
  echo "== abl=$abl rc=$?"; python tools/bench_brief.py < gpurun_out/r04_scale_nt$abl.json.log | cut -c1-260
done 2>&1 | tee gpurun_out/r04_scale_nt_stores.log
