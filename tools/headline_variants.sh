#!/bin/bash
# headline (ML-20M shape, k = 64) under the profile build's resident-regime switches: bins = 256 x MIN_ROUNDS, waves per bin,
# LDS floor per workgroup (82 = one workgroup per CU), triplets in flight.   tools/headline_variants.sh "<rounds> <waves> <excl_kb> <unr>" ...
cd "$(dirname "$0")/.."
export CORNAC_HIP_PROFILE=1
for v in "$@"; do
    set -- $v
    CORNAC_HIP_LDSBIN_MIN_ROUNDS=$1 CORNAC_HIP_LDSBIN_RES_WAVES=$2 CORNAC_HIP_LDSBIN_EXCL_KB=$3 CORNAC_HIP_LDSBIN_UNR=$4 \
      timeout 150 python bench.py --no-legs --no-rank --cpu-baseline-seconds 0 --steps 20 --warmup 3 2>/dev/null | grep '^{' | \
      python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', round(j['ms_per_step'],3), round(j['roofline']['frac'],4), j['train_stats'])" 2>&1 | tail -1
done
