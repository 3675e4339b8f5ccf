#!/bin/bash
# kernel timeline of the replicated-table regime on one rank (bench.py --force-dist): where the per-exchange cost goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_dist
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python $R/bench.py --force-dist --steps 4 --warmup 2 --no-rank --no-legs --cpu-baseline-seconds 0 ${EXTRA} > $O/bench.log 2>&1
cd $R
grep '^{' $O/bench.log | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline'].get('kernel'), j['roofline'].get('launches'), j['roofline'].get('avg_launch_ms'))"
python tools/rocpd_summary.py stats $O/t/t_results.db | head -12 | cut -c1-160
python tools/rocpd_summary.py timeline $O/t/t_results.db ${LAST:-110} > gpurun_out/dist_timeline.csv
tail -${LAST:-110} gpurun_out/dist_timeline.csv | cut -c1-130
