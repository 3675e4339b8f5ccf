#!/bin/bash
# kernel-trace stats of the binned path's arms (run through gpurun): gpurun_out/r02_binned_<arm>_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_binned
mkdir -p $O
for arm in "$@"; do
  rocprofv3 --kernel-trace --stats -d $O/$arm -o p -- python $R/tools/exp_binned.py --arms $arm --epochs 3 > $O/$arm.log 2>&1
  python $R/tools/rocpd_summary.py stats $O/$arm/p_results.db > $R/gpurun_out/r02_binned_${arm}_stats.csv
  tail -1 $O/$arm.log
  head -5 $R/gpurun_out/r02_binned_${arm}_stats.csv
done
