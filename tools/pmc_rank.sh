cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/bench_rank.py
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/rk1 -o p -- python $R/tools/bench_rank.py --repeats 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_MFMA -d $R/gpurun_out/rk2 -o p -- python $R/tools/bench_rank.py --repeats 1 > /dev/null 2>&1
cd $R; for d in rk1 rk2; do python tools/rocpd_summary.py pmc gpurun_out/$d/p_results.db 2>&1 | grep -i "rank_fused" | head -12; done
