"""`python bench.py --gpus N` must launch its own ranks (the driver's plain command line has no torchrun in front):
exercised on CPU with the gloo backend and a stand-in step (`--dry-run-cpu`): rendezvous on 127.0.0.1, barrier-bracketed
timing, MAX over ranks, one JSON line from rank 0 with n_gpus = N."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_single_rank_needs_no_launcher():
    d = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--dry-run-cpu"])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0


def test_gpus_2_spawns_two_ranks_and_reports_the_slowest():
    d = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run-cpu"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    # rank 1's stand-in step sleeps twice as long as rank 0's: the line must carry the slower rank's time
    assert d["ms_per_step"] >= 3.5, d
    assert abs(d["value"] - 1000.0 * 4 * 2 / (d["ms_per_step"] * 4 / 1e3)) < 1e-6 * d["value"]
