#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   kernel-trace stats of the default bench, PMC passes of the rank kernel (counters in their own runs).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/bench -o b -- python $R/bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 > $O/bench.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/rank1 -o p -- python $R/tools/bench_rank.py --repeats 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $O/rank2 -o p -- python $R/tools/bench_rank.py --repeats 1 > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py stats $O/bench/b_results.db > gpurun_out/r01b_bench_kernel_stats.csv
for d in rank1 rank2; do python tools/rocpd_summary.py pmc $O/$d/p_results.db | grep -E "^kernel|rank_fused"; done > gpurun_out/r01b_rank_pmc.csv
tail -1 $O/bench.log > gpurun_out/r01b_bench.json
