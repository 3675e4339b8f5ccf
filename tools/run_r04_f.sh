#!/bin/bash
# round 4, call 2: the VBPR look-ahead pipeline (parity tests, A/B against the round-3 order, per-kernel view), the
# 'align' rule kernels, the one-rank driver tax leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vbpr_gpu.py tests/test_sharded_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "vbpr or table_delta or sharded_mf or model_level" 2>&1 | grep -v amdgpu.ids | tail -8
for v in "X=0" "CORNAC_HIP_PROFILE=1" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_NO_LOOKAHEAD=1" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=6" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=8" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_VBPR_SWEEP_WGS=5"; do
  echo "== vbpr $v"; env $v timeout 300 python tools/bench_vbpr.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
done > gpurun_out/r04_vbpr_ab.log 2>&1
cat gpurun_out/r04_vbpr_ab.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vbpr -o v -- python $GRAFT_REPO_ROOT/tools/bench_vbpr.py > $GRAFT_REPO_ROOT/gpurun_out/prof_vbpr.log 2>&1 )
python tools/rocpd_summary.py stats gpurun_out/prof_vbpr/v_results.db > gpurun_out/r04_vbpr_kernel_stats.csv 2>&1; head -14 gpurun_out/r04_vbpr_kernel_stats.csv | cut -c1-180
timeout 900 python bench.py --steps 5 --warmup 1 --no-rank --cpu-baseline-seconds 0 --legs vbpr_tradesy,dist_tax > gpurun_out/r04_bench_legs_b.json.log 2> gpurun_out/r04_bench_legs_b.err
echo "bench rc=$?"; tail -3 gpurun_out/r04_bench_legs_b.err | cut -c1-300
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_legs_b.json.log") if l.startswith("{")][-1])
for n, l in j.get("legs", {}).items():
    print(n, json.dumps({k: v for k, v in l.items() if k in ("value", "ms_per_step", "error", "ml20m", "scale")})[:900], (l.get("roofline") or {}).get("frac"))
PY
