#!/bin/bash
# address-translation and fabric-read counters of the k = 128 scale leg's kernel (bpr_strata_kernel on an
# 11.5 GB random-access working set), beside the box calibration the leg prints.  Counters in their own runs.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_scale_tlb
mkdir -p $O
CMD="python $R/bench.py --config scale --steps 2 --warmup 1 --cpu-baseline-seconds 0"
run() { timeout 900 rocprofv3 --kernel-trace --pmc "${@:2}" -d $O/$1 -o p -- $CMD > $O/$1.log 2>&1; }
run t1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum
run t2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE
cd $R
for p in t1 t2; do python tools/rocpd_summary.py pmc $O/$p/p_results.db | grep -E "^kernel|strata_kernel"; done > gpurun_out/${ROUND:-r04}_scale_tlb_pmc.csv
cut -c1-200 gpurun_out/${ROUND:-r04}_scale_tlb_pmc.csv
grep '^{' $O/t1.log | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline'].get('avg_launch_ms'), j['config'].get('box'))"
