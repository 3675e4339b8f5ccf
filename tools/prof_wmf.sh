#!/bin/bash
# per-kernel durations of the WMF step (tools/wmf_ablate.py at the Netflix user count): rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_wmf_stats
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/b -o p -- python $R/tools/wmf_ablate.py --masks 0 > $O/run.log 2>&1
cd $R
python tools/rocpd_summary.py stats $O/b/p_results.db | sed 's/(float const.*)"/"/; s/(.*)"/"/' > gpurun_out/${ROUND:-r06}_wmf_kernel_stats.csv
cat gpurun_out/${ROUND:-r06}_wmf_kernel_stats.csv
tail -3 $O/run.log
