#!/bin/bash
# round 4, final call: full -m gpu suite, the default bench line, rocprofv3 kernel stats of the same line (without the
# dist_tax leg), counter passes (headline kernel, VBPR traffic, scale-leg translation counters, WMF), exchange-rule emulation
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/run_suite.sh
cp gpurun_out/bench.json gpurun_out/r04_bench_default.json.log
ROUND=r04 LEGS=vbpr_tradesy BENCH_ARGS="--legs mf_netflix,wmf_netflix,vbpr_tradesy,bpr_k128_scale" bash tools/profile_round.sh 2>&1 | tail -40
ROUND=r04 bash tools/pmc_scale_tlb.sh 2>&1 | tail -12
ROUND=r04 bash tools/pmc_wmf.sh 2>&1 | tail -16
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/emulate_ranks.py --ranks 4 --epochs 6 --grid "sqrt:4,8,16;align:4,8,16" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_emulate_ranks_bpr_r4.log
