#!/bin/bash
# per-dispatch durations of the ranking kernels inside the default bench (trained model, exclusion bitmap, merge)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_rank
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/b -o p -- python $R/bench.py --steps 20 --warmup 5 --no-legs --cpu-baseline-seconds 0 > $O/bench.log 2>&1
cd $R
python tools/rocpd_summary.py dispatches $O/b/p_results.db "%rank%" > gpurun_out/${ROUND:-r06}_rank_dispatches.csv
python tools/rocpd_summary.py dispatches $O/b/p_results.db "%excl_bitmap%" >> gpurun_out/${ROUND:-r06}_rank_dispatches.csv
python tools/rocpd_summary.py dispatches $O/b/p_results.db "%full_sort%" >> gpurun_out/${ROUND:-r06}_rank_dispatches.csv
cat gpurun_out/${ROUND:-r06}_rank_dispatches.csv
tail -c 1500 $O/bench.log | head -c 1200
