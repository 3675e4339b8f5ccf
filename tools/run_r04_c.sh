#!/bin/bash
# bias layout experiments of the strata kernel at the configs[4] slice (profile build)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" CORNAC_HIP_PROFILE=1 timeout 1200 python bench.py --config scale --steps 3 --warmup 1 --cpu-baseline-seconds 0 --flags ${FLAGS:-0} > gpurun_out/r04_scale_$tag.json.log 2> gpurun_out/r04_scale_$tag.err; echo "== $tag rc=$?"; python tools/bench_brief.py < gpurun_out/r04_scale_$tag.json.log; }
FLAGS=0 run base X=0
FLAGS=$((32 << 8)) run fullline X=0
FLAGS=0 run dense CORNAC_HIP_STRATA_DENSE_BIAS=1
FLAGS=$((8 << 8)) run nobias X=0
