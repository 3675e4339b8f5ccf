#!/bin/bash
# round 4, call 7: packed item records (row + bias line) of the strata form inside fit_epochs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bpr_gpu.py tests/test_sharded_gpu.py -x -q -m gpu -k "strata or forms or sharded_trainer or owned" 2>&1 | grep -v amdgpu.ids | tail -5
timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "invariants" 2>&1 | grep -v amdgpu.ids | tail -3
for v in "X=0" "CORNAC_HIP_PROFILE=1 CORNAC_HIP_STRATA_NO_PACK=1" "X=1"; do
  echo "== scale leg $v"
  env $v timeout 600 python bench.py --steps 3 --warmup 1 --no-rank --cpu-baseline-seconds 0 --legs bpr_k128_scale > gpurun_out/r04_scale_packed.json.log 2> gpurun_out/r04_scale_packed.err
  python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_scale_packed.json.log") if l.startswith("{")][-1])
l = j["legs"]["bpr_k128_scale"]
print(l.get("error") or (l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["frac_kernel_only"], l["roofline"]["avg_launch_ms"], l["train_stats"], l["box"].get("row_gather_512B_GBps")))
PY
done 2>&1 | tee gpurun_out/r04_scale_packed.log
timeout 600 python bench.py --config scale --steps 3 --warmup 1 --cpu-baseline-seconds 0 2>/dev/null | python tools/bench_brief.py | cut -c1-250
