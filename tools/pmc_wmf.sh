#!/bin/bash
# hardware counters of the fused WMF user-step kernel (tools/bench_wmf.py at the Netflix user count), counters in their own runs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_wmf
mkdir -p $O
CMD="python $R/tools/bench_wmf.py --shape 480189,2000,3000000 --k 128 --steps 16"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/w1 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD -d $O/w2 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/w3 -o p -- $CMD > /dev/null 2>&1
cd $R
for p in w1 w2 w3; do python tools/rocpd_summary.py pmc $O/$p/p_results.db | grep -E "^kernel|wmf_user_step"; done | sed 's/(float const.*)"/"/; s/(.*)"/"/' > gpurun_out/${ROUND:-r06}_wmf_pmc.csv
cat gpurun_out/${ROUND:-r06}_wmf_pmc.csv
