#!/bin/bash
# round 4, call 13: the resident-exchange tests (all parameter cases) and the model-level sharded fit through them
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_gpu.py -q -m gpu -k "resident or model_level or sharded_trainer" --timeout 200 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert|Timeout|^E " | cut -c1-400 | tail -24
