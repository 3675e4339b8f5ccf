#!/bin/bash
# one parametrised GPU call script (round 5): tools/run_r05.sh <tag> '<command>' — runs the command from the repo root
# with TMPDIR=/tmp, logs to gpurun_out/r05_<tag>.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=$1; shift
bash -c "$*" > gpurun_out/r05_$tag.log 2>&1
echo "rc=$?" >> gpurun_out/r05_$tag.log
tail -40 gpurun_out/r05_$tag.log
