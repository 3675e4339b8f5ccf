#!/bin/bash
# round 4, call 10: where the scale-shape driver's time goes with the records kept between chunks (kernel trace)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_gpu.py -x -q -m gpu -k "packed_records" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert" | cut -c1-300 | tail -6
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_scale_dist -o d -- python $GRAFT_REPO_ROOT/bench.py --config scale --force-dist --steps 3 --warmup 1 --cpu-baseline-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_scale_dist.log 2>&1 )
python tools/rocpd_summary.py stats gpurun_out/prof_scale_dist/d_results.db > gpurun_out/r04_scale_dist_kernel_stats.csv 2>&1; head -16 gpurun_out/r04_scale_dist_kernel_stats.csv | cut -c1-90,200-260
grep '^{' gpurun_out/prof_scale_dist.log | tail -1 | python tools/bench_brief.py | cut -c1-250
