#!/bin/bash
# one parametrised GPU call script (rounds 5-6): tools/run_gpu.sh <tag> '<command>' — runs the command from the repo root
# with TMPDIR=/tmp, logs to gpurun_out/r06_<tag>.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=$1; shift
bash -c "$*" > gpurun_out/r06_$tag.log 2>&1
echo "rc=$?" >> gpurun_out/r06_$tag.log
tail -40 gpurun_out/r06_$tag.log
