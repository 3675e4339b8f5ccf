#!/bin/bash
# round 4: the duty rows in groups of 4 (loads in flight together): resident tests, then the ML-20M one-rank tax twice
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_sharded_gpu.py -q -m gpu -k "resident" --timeout 60 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert |Timeout|^E " | cut -c1-300 | tail -12
for i in 1 2; do
CORNAC_BENCH_DIST_TAX_SHAPES=ml20m timeout 100 python bench.py --steps 3 --warmup 1 --no-rank --cpu-baseline-seconds 0 --legs dist_tax 2>/dev/null > gpurun_out/r04_bench_tax_resident_g$i.json.log
python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_tax_resident_g$i.json.log") if l.startswith("{")][-1])
l = j["legs"]["dist_tax"]["ml20m"]
print("plain %.3f driver %.3f tax %.4f %s" % (l["plain_ms_per_epoch"], l["driver_ms_per_epoch"], l["tax"], l["protocol"]))
PY
done
