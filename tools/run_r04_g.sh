#!/bin/bash
# round 4, call 3: VBPR with the batch rows' update on its own stream, vectorised table passes, the BPR exchange rule
# emulation (sqrt vs align), the Netflix-shape MF gate against the reference's threads
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vbpr_gpu.py tests/test_sharded_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -s -k "vbpr or table_delta or sharded_mf or reference_threads" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Netflix-shape|Error|error|assert" | tail -12
for v in "X=0" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=5" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=7" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=8" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_ROWS_ON_MAIN=1"; do
  echo "== vbpr $v"; env $v timeout 300 python tools/bench_vbpr.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-330
done > gpurun_out/r04_vbpr_ab2.log 2>&1
cat gpurun_out/r04_vbpr_ab2.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vbpr2 -o v -- python $GRAFT_REPO_ROOT/tools/bench_vbpr.py > $GRAFT_REPO_ROOT/gpurun_out/prof_vbpr2.log 2>&1 )
python tools/rocpd_summary.py stats gpurun_out/prof_vbpr2/v_results.db > gpurun_out/r04_vbpr_kernel_stats2.csv 2>&1; head -9 gpurun_out/r04_vbpr_kernel_stats2.csv | cut -c1-60,150-260
timeout 600 python tools/emulate_ranks.py --ranks 8 --epochs 6 --grid "sqrt:4,8,16;align:1,2,4,8,16" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_emulate_ranks_bpr.log
timeout 900 python bench.py --steps 5 --warmup 1 --no-rank --cpu-baseline-seconds 0 --legs dist_tax > gpurun_out/r04_bench_legs_c.json.log 2> gpurun_out/r04_bench_legs_c.err
echo "bench rc=$?"; tail -2 gpurun_out/r04_bench_legs_c.err | cut -c1-300
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_legs_c.json.log") if l.startswith("{")][-1])
for n, l in j.get("legs", {}).items():
    print(n, json.dumps({k: v for k, v in l.items() if k in ("value", "ms_per_step", "error", "ml20m", "scale")})[:900])
PY
