cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_vbpr_gpu.py -x -q -m gpu > gpurun_out/v_tests.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/v_tests.log | head -5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_vbpr -o v -- python $R/bench.py --legs vbpr_tradesy --no-rank --steps 2 --warmup 1 --cpu-baseline-seconds 0 > $R/gpurun_out/v_bench.log 2>$R/gpurun_out/v_bench.err
cd $R
python - <<'PY'
import sqlite3, json
cur=sqlite3.connect('gpurun_out/prof_vbpr/v_results.db').cursor()
rows=list(cur.execute("select name,start,end,queue_id,stream_id from kernels order by start"))
idx=[k for k,r in enumerate(rows) if 'adam_sweep' in r[0]]
k0=idx[2000]; t0=rows[k0][1]
for r in rows[k0-6:k0+12]:
    print("%-34s start %8.1f  end %8.1f dur %6.1f  q%s"%(r[0].split('(')[0][-34:],(r[1]-t0)/1e3,(r[2]-t0)/1e3,(r[2]-r[1])/1e3,r[3]))
for l in open('gpurun_out/v_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); v=d['legs']['vbpr_tradesy']; print(v.get('ms_per_step'), v.get('roofline',{}).get('frac'), v.get('error'))
PY
