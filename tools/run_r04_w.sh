#!/bin/bash
# round 4: bench.py's own multi-GPU step on one rank through RCCL (what `--gpus N` runs per rank): ML-20M shape, resident exchange
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python bench.py --force-dist --steps 10 --warmup 3 --no-legs --no-rank --cpu-baseline-seconds 0 2> gpurun_out/r04_bench_forcedist.err | tee gpurun_out/r04_bench_forcedist.json.log | python tools/bench_brief.py | cut -c1-400
tail -3 gpurun_out/r04_bench_forcedist.err | cut -c1-300
