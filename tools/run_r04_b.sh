#!/bin/bash
# ablations of the strata kernel at the configs[4] slice (profile build): bit0 no membership test, bit1 no stores,
# bit2 no row loads, bit3 no bias
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for abl in ${ABLS:-0 8 1 9 2 4}; do
  CORNAC_HIP_PROFILE=1 timeout 1200 python bench.py --config scale --steps 3 --warmup 1 --cpu-baseline-seconds 0 --flags $((abl << 8)) > gpurun_out/r04_scale_abl$abl.json.log 2> gpurun_out/r04_scale_abl$abl.err
  echo "== abl=$abl rc=$?"
  python tools/bench_brief.py < gpurun_out/r04_scale_abl$abl.json.log
done
