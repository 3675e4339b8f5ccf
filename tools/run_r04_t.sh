#!/bin/bash
# round 4, last call: the resident-exchange tests as shipped and the default bench line (all legs, dist_tax with the
# resident exchange and the exchange schedule)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 330 python bench.py > gpurun_out/r04_bench_default_final.json.log 2> gpurun_out/r04_bench_default_final.err
echo "bench rc=$? lines=$(wc -l < gpurun_out/r04_bench_default_final.json.log)"
python tools/bench_brief.py < gpurun_out/r04_bench_default_final.json.log 2>&1 | cut -c1-400 | head -30
timeout 200 python -m pytest tests/test_sharded_gpu.py -q -m gpu -k "resident" --timeout 90 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert|Timeout|^E " | cut -c1-300 | tail -8
