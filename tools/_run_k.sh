cd $GRAFT_REPO_ROOT
timeout 240 python tools/bench_sharded.py --users 1000000 --items 500000 --degree 20 --k 128 --micro-batch 2000000 --trace > gpurun_out/sharded_trace.log 2>&1; grep -v -i "rccl\|hostname\|version\|amdgpu" gpurun_out/sharded_trace.log | sed -n 30,75p
