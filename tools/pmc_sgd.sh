#!/bin/bash
# Hardware-counter passes of the PRODUCTION hogwild SGD kernel (bench.py's default launch), counters in their own runs
# (no trace domains besides --kernel-trace): atomics executed by the L2s and forwarded to the fabric (EA), tag / write
# request stalls, wave stall breakdown.  Summaries: gpurun_out/r02_sgd_pmc.csv (copied to profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_sgd
mkdir -p $O
CMD="python $R/bench.py --steps 3 --warmup 1 --no-rank --cpu-baseline-seconds 0"
run() { rocprofv3 --kernel-trace --pmc "${@:2}" -d $O/$1 -o p -- $CMD > $O/$1.log 2>&1; }
run p1 TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum GRBM_GUI_ACTIVE
run p2 TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum TCC_BUSY_sum GRBM_GUI_ACTIVE
run p3 TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
run p4 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY
cd $R
for p in p1 p2 p3 p4; do python tools/rocpd_summary.py pmc $O/$p/p_results.db | grep -E "^kernel|hogwild"; done > gpurun_out/r02_sgd_pmc.csv
cat gpurun_out/r02_sgd_pmc.csv
