#!/bin/bash
# Counters of the configs[4] slice's kernel (round 5: bpr_ldsbin_kernel<2,4> with passing bins): atomics received by the L2s /
# forwarded to the fabric, fabric read / write requests, FETCH_SIZE / WRITE_SIZE (each in its own run, only --kernel-trace
# beside the counters), beside the line the run prints.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r05}
O=$R/gpurun_out/prof_scale_$ROUND
mkdir -p $O
CMD="python $R/bench.py --config scale --steps 2 --warmup 1 --cpu-baseline-seconds 0"
run() { timeout 900 rocprofv3 --kernel-trace --pmc "${@:2}" -d $O/$1 -o p -- $CMD > $O/$1.log 2>&1; }
run s1 TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum GRBM_GUI_ACTIVE
run s2 TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
run s3 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
run s4 FETCH_SIZE
run s5 WRITE_SIZE
cd $R
for p in s1 s2 s3 s4 s5; do python tools/rocpd_summary.py pmc $O/$p/p_results.db | grep -E "^kernel|ldsbin_kernel|strata_kernel"; done > gpurun_out/${ROUND}_scale_pmc.csv
cut -c1-220 gpurun_out/${ROUND}_scale_pmc.csv
grep '^{' $O/s1.log | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline'])"
