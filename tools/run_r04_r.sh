#!/bin/bash
# round 4, call 17: resident + records(rule) tests, the one-rank driver tax at both shapes with the exchange schedule
# (ML-20M: 16 per epoch inside one launch, sqrt; configs[4] slice: one exchange every 4 epochs, align)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_gpu.py -q -m gpu -k "resident or packed_records or model_level" --timeout 300 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert|Timeout|^E " | cut -c1-400 | tail -16
timeout 600 python bench.py --steps 5 --warmup 1 --no-rank --cpu-baseline-seconds 0 --legs dist_tax > gpurun_out/r04_bench_legs_f.json.log 2> gpurun_out/r04_bench_legs_f.err
echo "bench rc=$?"; tail -2 gpurun_out/r04_bench_legs_f.err | cut -c1-300
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_legs_f.json.log") if l.startswith("{")][-1])
print("headline ms", j["ms_per_step"], "frac", j["roofline"]["frac"])
for n, l in j.get("legs", {}).items():
    print(n, json.dumps({k: v for k, v in l.items() if k in ("value", "error", "ml20m", "scale")})[:1400])
PY
