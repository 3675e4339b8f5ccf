#!/bin/bash
# per-workload kernel stats of the MF epoch at the Netflix shape: one rocprofv3 --kernel-trace --stats run per item skew
# (bench.py's Zipf 0.45 and SURVEY 8d's 0.8), so that mf_blocks_kernel's average duration is not a mix of the two
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r06}
for z in 0.45 0.8; do
  O=$R/gpurun_out/prof_mf_zipf_$z
  mkdir -p $O
  timeout 600 rocprofv3 --kernel-trace --stats -d $O -o p -- python $R/tools/mf_zipf.py $z > $O/run.log 2>&1
  ( cd $R && echo "# zipf $z" && python tools/rocpd_summary.py stats $O/p_results.db | sed 's/(.*)"/"/' | head -8 && grep -v "^W\|^$" $O/run.log | tail -3 )
done > $R/gpurun_out/${ROUND}_mf_zipf_kernel_stats.csv
cat $R/gpurun_out/${ROUND}_mf_zipf_kernel_stats.csv
