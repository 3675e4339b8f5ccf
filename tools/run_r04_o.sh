#!/bin/bash
# round 4, call 12: the resident exchange (one launch per epoch, exchange points inside it): accounting tests, one rank
# through RCCL, and the one-rank driver tax at the ML-20M shape with it and with chunk launches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_sharded_gpu.py -x -q -m gpu -q -k "resident" --timeout 150 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert|Timeout" | cut -c1-400 | tail -14
for mode in resident chunks; do
  if [ $mode = chunks ]; then export CORNAC_BENCH_DIST_CHUNKS=1; else unset CORNAC_BENCH_DIST_CHUNKS; fi
  CORNAC_BENCH_DIST_TAX_SHAPES=ml20m timeout 300 python bench.py --steps 5 --warmup 1 --no-rank --cpu-baseline-seconds 0 --legs dist_tax > gpurun_out/r04_bench_tax_$mode.json.log 2> gpurun_out/r04_bench_tax_$mode.err
  echo "bench $mode rc=$?"; tail -2 gpurun_out/r04_bench_tax_$mode.err | cut -c1-300
  python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_tax_$mode.json.log") if l.startswith("{")][-1])
print("headline ms", j["ms_per_step"], "frac", j["roofline"]["frac"])
for n, l in j.get("legs", {}).items():
    print(n, json.dumps({k: v for k, v in l.items() if k in ("value", "error", "ml20m", "scale")})[:900])
PY
done
