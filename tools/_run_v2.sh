cd $GRAFT_REPO_ROOT
for cfg in "ONE 8" "ONE 8" "TWO 6" "TWO 6" "TWO 5" "TWO 7"; do
set -- $cfg
if [ $1 = ONE ]; then export CORNAC_HIP_VBPR_ONE_STREAM=1; else unset CORNAC_HIP_VBPR_ONE_STREAM; fi
CORNAC_HIP_VBPR_SWEEP_WGS=$2 timeout 300 python bench.py --legs vbpr_tradesy --no-rank --steps 2 --warmup 1 --cpu-baseline-seconds 0 > gpurun_out/v_bench.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/v_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); v=d['legs']['vbpr_tradesy']; print("$cfg", v.get('ms_per_step'), v.get('roofline',{}).get('frac'), v.get('error'))
PY
done
