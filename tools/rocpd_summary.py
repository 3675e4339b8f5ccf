#!/usr/bin/env python3
"""Turns rocprofv3 (ROCm 7.2, rocpd sqlite output) results into small text summaries for profiles/.

    python tools/rocpd_summary.py stats  gpurun_out/prof/stats/r01_results.db      > profiles/r01_kernel_stats.csv
    python tools/rocpd_summary.py pmc    gpurun_out/prof/pmc_fetch/r01_results.db  > profiles/r01_pmc_fetch.csv
"""
import sqlite3
import sys


def stats(path):
    cur = sqlite3.connect(path).cursor()
    print("kernel,calls,total_us,avg_us,percent")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print('"%s",%d,%.1f,%.1f,%.2f' % (name, calls, total, avg, pct))


def pmc(path):
    cur = sqlite3.connect(path).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
         "from counters_collection group by kernel_name, counter_name order by avg(duration)*count(*) desc")
    print("kernel,counter,dispatches,avg_value,min_value,max_value,avg_duration_ns")
    for r in cur.execute(q):
        print('"%s",%s,%d,%.1f,%.1f,%.1f,%.0f' % r)


def dispatches(path, pattern="%"):
    """every dispatch of the kernels whose name matches the SQL LIKE pattern, in launch order"""
    cur = sqlite3.connect(path).cursor()
    print("kernel,grid_x,workgroup_x,lds_bytes,vgprs,duration_us")
    for name, gx, wx, lds, slds, vg, dur in cur.execute(
            "select name, grid_x, workgroup_x, lds_size, static_lds_size, vgpr_count, duration from kernels "
            "where name like ? order by start", (pattern,)):
        print('"%s",%d,%d,%d,%d,%.1f' % (name[:90], gx, wx, lds + slds, vg, dur / 1e3))


def timeline(path, last="120"):
    """the last N dispatches in start order: offset from the first of them, duration, gap since the previous kernel's end"""
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, duration from kernels order by start"))[-int(last):]
    print("kernel,start_us,duration_us,gap_before_us")
    t0, prev_end = rows[0][1], rows[0][1]
    for name, start, dur in rows:
        print('"%s",%.1f,%.1f,%.1f' % (name[:60], (start - t0) / 1e3, dur / 1e3, (start - prev_end) / 1e3))
        prev_end = max(prev_end, start + dur)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "dispatches": dispatches, "timeline": timeline}[sys.argv[1]](*sys.argv[2:])
