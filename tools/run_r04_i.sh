#!/bin/bash
# round 4, call 5: MF deterministic dataflow kernel through a cooperative launch, block-rotation probe, rank kernel with the
# exclusion words requested ahead of the MFMA chain
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mf_gpu.py tests/test_score_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -s -k "mf_netflix_shape_deterministic or mf_full_size" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Netflix-shape|rror" | cut -c1-400 | tail -6
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 1 --no-legs --cpu-baseline-seconds 0 > gpurun_out/r04_bench_rank_d$i.json.log 2> gpurun_out/r04_bench_rank_d$i.err
python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_rank_d$i.json.log") if l.startswith("{")][-1])
r = j["rank"]; print("headline", j["ms_per_step"], j["roofline"]["frac"], "rank ms", r["ms"], r["device_ms"], r["ms_no_exclusions_device_only"], r["roofline"])
PY
done
