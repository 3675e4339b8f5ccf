#!/usr/bin/env python3
"""prints the key numbers of bench.py JSON lines (one per line) read from the files named on the command line, or from stdin"""
import json
import sys

import fileinput

for line in fileinput.input():
    if not line.startswith("{"):
        continue
    j = json.loads(line)
    r = j.get("roofline") or {}
    print("%s n_gpus %s value %.4g %s ms/step %.3f | form %s par %s | kernel %s launches %s avg %.3f ms frac %s" % (
        j["metric"], j["n_gpus"], j["value"], j["unit"], j["ms_per_step"], j["config"].get("form"), j["config"].get("parallelism"),
        r.get("kernel"), r.get("launches"), r.get("avg_launch_ms") or 0, r.get("frac")))
    for n, l in (j.get("legs") or {}).items():
        rr = l.get("roofline") or {}
        print("   leg %-16s value %.4g ms/step %.4g frac %s kernel %s %s" % (n, l.get("value") or 0, l.get("ms_per_step") or 0,
              rr.get("frac"), rr.get("kernel"), l.get("error") or ""))
    if "rank" in j:
        print("   rank ms %.3f frac %.3f" % (j["rank"]["ms"], j["rank"]["roofline"]["frac"]))
