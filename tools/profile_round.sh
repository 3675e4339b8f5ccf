#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   kernel-trace stats of the DEFAULT bench command (every leg), whose average duration of the SGD kernel must agree
#   with the HIP-event figure in the JSON line of the same run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/bench -o b -- python $R/bench.py > $O/bench.log 2>&1
cd $R
python tools/rocpd_summary.py stats $O/bench/b_results.db > gpurun_out/r02_bench_kernel_stats.csv
grep '^{' $O/bench.log | tail -1 > gpurun_out/r02_bench_profiled.json.log
head -12 gpurun_out/r02_bench_kernel_stats.csv
