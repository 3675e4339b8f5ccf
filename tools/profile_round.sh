#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun); ROUND names the files.
#   1. kernel-trace stats of the DEFAULT bench command (every leg): the SGD kernel's average duration must agree with the
#      HIP-event figure in the JSON line of the same run;
#   2. hardware-counter passes of the headline SGD kernel (counters in their own runs, only --kernel-trace beside them):
#      atomics received by the L2s / forwarded to the fabric, stalls, wave-state breakdown;
#   3. FETCH_SIZE / WRITE_SIZE passes (separate runs) over the headline kernel and the legs' kernels.
ROUND=${ROUND:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$ROUND
mkdir -p $O
if [ "${SKIP_STATS:-0}" != 1 ]; then
# (BENCH_ARGS: round 4 profiles the default line without the dist_tax leg — an RCCL process group under the tracer —
# `--legs mf_netflix,wmf_netflix,vbpr_tradesy,bpr_k128_scale`; every kernel of the line is in that run)
rocprofv3 --kernel-trace --stats -d $O/bench -o b -- python $R/bench.py $BENCH_ARGS > $O/bench.log 2>&1
( cd $R && python tools/rocpd_summary.py stats $O/bench/b_results.db > gpurun_out/${ROUND}_bench_kernel_stats.csv
  grep '^{' $O/bench.log | tail -1 > gpurun_out/${ROUND}_bench_profiled.json.log
  head -14 gpurun_out/${ROUND}_bench_kernel_stats.csv | cut -c1-200 )
fi
KPAT="ldsbin|strata_kernel|bpr_hogwild"
CMD="python $R/bench.py --steps 3 --warmup 1 --no-rank --cpu-baseline-seconds 0 --no-legs"
run() { timeout 600 rocprofv3 --kernel-trace --pmc "${@:2}" -d $O/$1 -o p -- $CMD > $O/$1.log 2>&1; }
run p1 TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum GRBM_GUI_ACTIVE
run p2 TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum TCC_BUSY_sum GRBM_GUI_ACTIVE
run p3 TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
run p4 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY
run p5 SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES
( cd $R && for p in p1 p2 p3 p4 p5; do python tools/rocpd_summary.py pmc $O/$p/p_results.db | grep -E "^kernel|$KPAT"; done > gpurun_out/${ROUND}_sgd_pmc.csv
  cut -c1-200 gpurun_out/${ROUND}_sgd_pmc.csv )
if [ "${SKIP_LEGS:-0}" != 1 ]; then
CMD="env CORNAC_BENCH_VBPR_FEEDBACK=30000 python $R/bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 0 --legs ${LEGS:-mf_netflix,wmf_netflix,vbpr_tradesy} --rank-full-users 0"
run fetch FETCH_SIZE
run write WRITE_SIZE
( cd $R && for p in fetch write; do python tools/rocpd_summary.py pmc $O/$p/p_results.db | grep -E "^kernel|mf_hogwild|mf_blocks|mf_det_chain|wmf_user_step|adam_sweep|rank_fused|feat_adam|touched|vbpr_|$KPAT"; done > gpurun_out/${ROUND}_legs_pmc.csv
  cut -c1-220 gpurun_out/${ROUND}_legs_pmc.csv )
fi
