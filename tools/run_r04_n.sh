#!/bin/bash
# round 4, call 11: the driver-with-records test (aggregate comparison) and the one-rank driver tax leg with persistent
# exchange buffers and a warm-up of the timed region's own pattern
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_gpu.py -x -q -m gpu -k "packed_records or sharded_trainer or table_delta" --durations=3 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert|s call" | cut -c1-300 | tail -8
timeout 900 python bench.py --steps 5 --warmup 1 --no-rank --cpu-baseline-seconds 0 --legs dist_tax > gpurun_out/r04_bench_legs_e.json.log 2> gpurun_out/r04_bench_legs_e.err
echo "bench rc=$? lines=$(wc -l < gpurun_out/r04_bench_legs_e.json.log)"
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_legs_e.json.log") if l.startswith("{")][-1])
for n, l in j.get("legs", {}).items():
    print(n, json.dumps({k: v for k, v in l.items() if k in ("value", "error", "ml20m", "scale")})[:900])
PY
