#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output of `rocprofv3 --kernel-trace --stats`) as text:
per-kernel calls / total / average / min / max duration plus launch geometry and register/LDS footprint.

    python tools/rocpd_summary.py gpurun_out/prof_r1/bench_results.db > profiles/r1_bench_kernel_stats.txt
"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
print("# source: %s" % sys.argv[1])
print("# per-kernel statistics (durations in microseconds)")
print("%-60s %6s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
rows = cur.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
    "order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
for name, n, s, a, mn, mx in rows:
    print("%-60s %6d %14.1f %12.1f %12.1f %12.1f %6.2f%%" % (name[:60], n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3,
                                                            100.0 * s / tot))
print()
print("# launch geometry / resources per kernel")
for r in cur.execute(
        "select distinct name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
        "from kernels"):
    print("%-60s grid=%d wg=%d lds=%d scratch=%d vgpr=%d agpr=%d sgpr=%d" % ((r[0][:60],) + tuple(r[1:])))
