#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bpr_gpu.py tests/test_sharded_gpu.py -x -q -m gpu -k "strata or owned or ldsbin or forms" 2>&1 | tail -4
for i in 1 2; do
timeout 1200 python bench.py --config scale --steps 4 --warmup 1 --cpu-baseline-seconds 0 > gpurun_out/r04_scale_plain$i.json.log 2> gpurun_out/r04_scale_plain$i.err
echo "== plain rc=$?"; python tools/bench_brief.py < gpurun_out/r04_scale_plain$i.json.log
done
python - <<'PY'
import sys; sys.path.insert(0, ".")
from cornac_amd import _lib
print(_lib.device_probe(0))
PY
tools/gather_probe 6144 | grep -E "rmw   dword    6|load  dword    6" | head -4
