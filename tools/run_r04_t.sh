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

# kernel view of the resident exchange (one rank through RCCL at the ML-20M shape): the EXCH instantiation of the LDS-bin kernel,
# the wait-counter / set-flag kernels of the communication stream, the flush
( cd /tmp && CORNAC_BENCH_DIST_TAX_SHAPES=ml20m timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_resident -o d -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-rank --cpu-baseline-seconds 0 --legs dist_tax > $GRAFT_REPO_ROOT/gpurun_out/prof_resident.log 2>&1 )
python tools/rocpd_summary.py stats gpurun_out/prof_resident/d_results.db > gpurun_out/r04_resident_kernel_stats.csv 2>&1; head -12 gpurun_out/r04_resident_kernel_stats.csv | cut -c1-110,200-260
