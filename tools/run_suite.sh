#!/bin/bash
# full -m gpu suite, then the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/suite.log
tail -30 gpurun_out/suite.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?"
tail -3 gpurun_out/bench.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/bench.json"))
print({k:j[k] for k in ("value","ms_per_step")}, j["roofline"], j.get("cpu_baseline"))
print(j["config"].get("form"), j["train_stats"])
for n,l in j.get("legs",{}).items(): print(n, {k:l.get(k) for k in ("value","ms_per_step","error")}, l.get("roofline",{}).get("frac"), l.get("roofline",{}).get("kernel"))
print(j["rank"]["ms"], j["rank"]["roofline"]["frac"])
PY
