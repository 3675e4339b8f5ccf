#!/bin/bash
# quick GPU check: a pytest -k selection, then optional extra command
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu -k "${K:-ldsbin}" > gpurun_out/quick_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/quick_tests.log
tail -${TAIL:-12} gpurun_out/quick_tests.log
if [ -n "$CMD" ]; then eval "$CMD"; fi
