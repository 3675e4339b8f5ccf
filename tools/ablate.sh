#!/bin/bash
# headline kernel (bpr_ldsbin_kernel<1,4>, ML-20M shape) with the profile build's ablation bits (hogwild_flags bits 8..: 1 no
# membership test, 2 no updates (LDS item rows + user atomics), 4 no user-row loads, 8 fresh load + write-through RMW instead of
# the atomics, 16 plain racy RMW, 32 no item-row update in the LDS, 128 item-row update without the row lock; CORNAC_HIP_LDSBIN_X
# bit 0 no cross-lane sum of the score, bit 1 no sigmoid): what each request stream costs.  CORNAC_HIP_LDSBIN_PRESAMPLE=1 in the
# environment: the same on the pre-sampled (training-only) launch.  Results of an ablated run are garbage, only its time counts.
cd "$(dirname "$0")/.."
export CORNAC_HIP_PROFILE=1
for bits in ${BITS:-0 1 2 4 3 6 5 7 16 17 32 128 144 48}; do
  f=$((bits * 256))
  timeout 300 python bench.py --steps 8 --warmup 2 --flags $f --cpu-baseline-seconds 0 --no-rank --no-legs 2>/dev/null | grep '^{' | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate bits %2d' % $bits, ' ms/epoch %.3f' % d['ms_per_step'], ' kernel ms %.3f' % d['roofline']['avg_launch_ms'], ' frac %.3f' % d['roofline']['frac'])"
done
