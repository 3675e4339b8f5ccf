cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "reference_threads or statistical or invariants" -s 2>&1 | grep -v amdgpu.ids | tail -12
for c in 0 1; do
  CORNAC_HIP_PROFILE=1 CORNAC_HIP_STRATA_CONTIG=$c timeout 1200 python bench.py --config scale --steps 4 --warmup 1 --cpu-baseline-seconds 0 > gpurun_out/r04_scale_contig$c.json.log 2> gpurun_out/r04_scale_contig$c.err
  echo "== contig=$c rc=$?"; tail -2 gpurun_out/r04_scale_contig$c.err | cut -c1-300
  python tools/bench_brief.py < gpurun_out/r04_scale_contig$c.json.log
done
