#!/bin/bash
# configs[4] through bench.py on one rank: plain, replicated-table regime (sparse exchange), row-sharded regime
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "strata or sharded or table_delta" > gpurun_out/quick_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/quick_tests.log; tail -4 gpurun_out/quick_tests.log
for v in "" "--force-dist --sync-per-epoch 8" "--force-dist --sync-per-epoch 64" "--force-dist --sharded-items"; do
  tag=$(echo "plain$v" | tr -d ' -')
  timeout 1200 python bench.py --config scale --steps 4 --warmup 1 --cpu-baseline-seconds 0 $v > gpurun_out/r03_scale_$tag.json.log 2> gpurun_out/r03_scale_$tag.err
  echo "== $v rc=$?"; tail -2 gpurun_out/r03_scale_$tag.err | cut -c1-300
  python tools/bench_brief.py < gpurun_out/r03_scale_$tag.json.log
done
