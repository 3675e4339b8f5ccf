#!/bin/bash
# round 4, call 9: packed records through the multi-GPU driver (chunk_records), the packed round-trip test, the
# ranking-metric A/B of the LDS-bin negatives, the one-rank driver tax with the records kept between chunks
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bpr_gpu.py tests/test_sharded_gpu.py -x -q -m gpu -s -k "packed or ranking_metrics or sharded_trainer or table_delta" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Recall|rror|assert" | cut -c1-300 | tail -12
timeout 900 python bench.py --steps 5 --warmup 1 --no-rank --cpu-baseline-seconds 0 --legs dist_tax > gpurun_out/r04_bench_legs_d.json.log 2> gpurun_out/r04_bench_legs_d.err
echo "bench rc=$? lines=$(wc -l < gpurun_out/r04_bench_legs_d.json.log)"; tail -2 gpurun_out/r04_bench_legs_d.err | cut -c1-300
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_legs_d.json.log") if l.startswith("{")][-1])
for n, l in j.get("legs", {}).items():
    print(n, json.dumps({k: v for k, v in l.items() if k in ("value", "error", "ml20m", "scale")})[:900])
PY
timeout 600 python bench.py --config scale --force-dist --steps 3 --warmup 1 --cpu-baseline-seconds 0 2>/dev/null | python tools/bench_brief.py | cut -c1-250
