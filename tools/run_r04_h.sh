#!/bin/bash
# round 4, call 4: VBPR sweep with two units per thread (workgroups per CU sweep), VEBPR float64, MF divergence report
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vbpr_gpu.py tests/test_bpr_gpu.py tests/test_mf_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "vbpr or vebpr or divergence or float64" 2>&1 | grep -v amdgpu.ids | tail -6
for v in "X=0" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=2" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=4" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=5" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=6"; do
  echo "== vbpr $v"; env $v timeout 300 python tools/bench_vbpr.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c100-330
done > gpurun_out/r04_vbpr_ab3.log 2>&1
cat gpurun_out/r04_vbpr_ab3.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vbpr3 -o v -- python $GRAFT_REPO_ROOT/tools/bench_vbpr.py > $GRAFT_REPO_ROOT/gpurun_out/prof_vbpr3.log 2>&1 )
python tools/rocpd_summary.py stats gpurun_out/prof_vbpr3/v_results.db > gpurun_out/r04_vbpr_kernel_stats3.csv 2>&1; head -9 gpurun_out/r04_vbpr_kernel_stats3.csv | cut -c1-44,150-260
