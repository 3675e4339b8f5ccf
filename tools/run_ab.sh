#!/bin/bash
# quick A/B: the ldsbin tests, the headline (no legs) and the one-rank replicated-table regime
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "${K:-ldsbin}" 2>&1 | tail -3
P='import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms"])'
for i in 1 2; do python bench.py --no-legs --no-rank --cpu-baseline-seconds 0 --steps 20 --warmup 3 2>/dev/null | python -c "$P" plain; done
python bench.py --force-dist --no-legs --no-rank --cpu-baseline-seconds 0 --steps 10 --warmup 3 2>/dev/null | python -c "$P" dist16
