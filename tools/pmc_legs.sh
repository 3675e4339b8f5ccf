#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes as MI355X_MICROARCH.md prescribes) of the dominant
# kernels of the bench legs and of the ranking kernel.  Summary: gpurun_out/r02_legs_pmc.csv (copied to profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_legs
mkdir -p $O
CMD="env CORNAC_HIP_VBPR_ONE_STREAM=1 CORNAC_BENCH_VBPR_FEEDBACK=30000 python $R/bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 0 --legs mf_netflix,wmf_netflix,vbpr_tradesy --rank-full-users 0"
run() { timeout 400 rocprofv3 --kernel-trace --pmc "${@:2}" -d $O/$1 -o p -- $CMD > $O/$1.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
for p in fetch write; do python tools/rocpd_summary.py pmc $O/$p/p_results.db | grep -E "^kernel|mf_hogwild|wmf_user_step|adam_sweep|rank_fused|bpr_hogwild|feat_adam|touched"; done > gpurun_out/r02_legs_pmc.csv
cat gpurun_out/r02_legs_pmc.csv | cut -c1-220
