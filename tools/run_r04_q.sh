#!/bin/bash
# round 4, call 15: diagnostic of the resident exchange's bookkeeping over several epochs
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 300 python tools/diag_resident.py 2>&1 | grep -v amdgpu.ids | tail -12

