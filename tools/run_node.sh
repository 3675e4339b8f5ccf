#!/bin/bash
# Node-readiness kit: ONE command that produces the 1 / 2 / 4 / 8-GPU tables of both multi-GPU paths on an MI355X node.
#
#   bash tools/run_node.sh [max_gpus=8] [out_dir=gpurun_out/node]
#
#   table 1  bench.py --gpus N                      headline shape (ML-20M, k = 64): regime 1, resident exchange
#   table 2  bench.py --config scale --gpus N --rings K   configs[4] shape (100 M x 10 M, k = 128): regime 2, the ring
#            conveyor of item blocks, K in {1, 2, 4} strided rings; per step: launch (compute) ms vs transfer ms from HIP
#            events on the compute and the communication stream (bench.py `per_step`)
#
# Every run is `python bench.py --gpus N ...`, which re-launches itself under torch.distributed.run with one rank per GPU
# (127.0.0.1 rendezvous, RCCL); each prints ONE JSON line on rank 0, kept under out_dir and summarised at the end.
# With one GPU only (the gpurun box) the tables have their N = 1 row and the conveyor is also laid out for 8 virtual ranks
# (`--force-dist --virtual-world 8`: a node rank's launch sizes and block copies on one device).
# Nothing here reads /root/reference; no CPU baseline legs (--cpu-baseline-seconds 0), no extra legs (--no-legs).
cd "$(dirname "$0")/.."
MAXG=${1:-8}
OUT=${2:-gpurun_out/node}
mkdir -p "$OUT"
export TMPDIR=${TMPDIR:-/tmp}
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python - <<'EOF'
import torch
print(torch.cuda.device_count())
EOF
)
echo "[run_node] $HAVE GPU(s) visible, tables up to N = $MAXG"
STEPS=${STEPS:-8}
run() {   # tag, timeout, args...
  local tag=$1 to=$2; shift 2
  echo "[run_node] $tag: python bench.py $*"
  timeout "$to" python bench.py "$@" > "$OUT/$tag.json.log" 2> "$OUT/$tag.err"
  echo "[run_node] $tag rc=$?" | tee -a "$OUT/$tag.err"
}
for N in 1 2 4 8; do
  [ "$N" -le "$MAXG" ] && [ "$N" -le "$HAVE" ] || continue
  run "ml20m_n$N" 900 --gpus $N --steps $STEPS --warmup 2 --no-legs --no-rank --cpu-baseline-seconds 0
  if [ "$N" -eq 1 ]; then
    run "scale_n1" 1500 --config scale --gpus 1 --steps $STEPS --warmup 3 --cpu-baseline-seconds 0 --no-legs
    for K in 1 2 4; do
      run "scale_v8_rings$K" 1500 --config scale --gpus 1 --force-dist --virtual-world 8 --rings $K --steps $STEPS --warmup 3 --cpu-baseline-seconds 0 --no-legs
    done
  else
    for K in 1 2 4; do
      [ $((2 * N * K)) -le 64 ] || continue
      run "scale_n${N}_rings$K" 1800 --config scale --gpus $N --rings $K --steps $STEPS --warmup 3 --cpu-baseline-seconds 0 --no-legs
    done
  fi
done
python - "$OUT" <<'EOF'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for p in sorted(glob.glob(os.path.join(out, "*.json.log"))):
    tag = os.path.basename(p)[:-9]
    line = None
    for l in open(p):
        if l.startswith('{"metric"'):
            line = l
    if not line:
        rows.append((tag, None))
        continue
    rows.append((tag, json.loads(line)))
base = {}
print("\n%-22s %6s %14s %12s %9s %9s %11s %11s  %s" % ("run", "gpus", "triplets/s", "ms/step", "eff", "frac", "launch ms", "transfer ms", "regime"))
for tag, d in rows:
    if d is None:
        print("%-22s   (no JSON line: see %s/%s.err)" % (tag, out, tag))
        continue
    fam = tag.split("_n")[0].split("_v")[0] + ("_rings" + tag.split("rings")[1] if "rings" in tag else "")
    n = d.get("n_gpus", 1)
    if n == 1 and "_v" not in tag:
        base.setdefault(tag.split("_")[0], d["value"])
    b = base.get(tag.split("_")[0])
    eff = d["value"] / (b * n) if b else float("nan")
    ps = d.get("per_step", {})
    print("%-22s %6d %14.4g %12.3f %9.3f %9.3f %11s %11s  %s" % (
        tag, n, d["value"], d["ms_per_step"], eff, d.get("roofline", {}).get("frac", float("nan")),
        ("%.3f" % ps["launch_ms"]) if "launch_ms" in ps else "-", ("%.3f" % ps["transfer_ms"]) if "transfer_ms" in ps else "-",
        str(d.get("config", {}).get("parallelism", ""))[:60]))
print("\n(eff = value / (N x the N = 1 value of the same shape); the driver computes its own from the raw lines under %s)" % out)
EOF
