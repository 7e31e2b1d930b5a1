#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats) from a rocpd sqlite file as text.
usage: python tools/prof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print(f"# source: {sys.argv[1]}  (rocprofv3 --kernel-trace --stats; durations in microseconds)")
print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name[:150]}")
try:
    rows = list(cur.execute("select name, min(duration), max(duration), avg(duration), count(*), max(vgpr_count), max(accum_vgpr_count), "
                            "max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc limit 12"))
    print("\n# per-kernel launch geometry / registers (duration in ns)")
    for r in rows:
        print(f"{r[0][:70]:70s} n={r[4]} min={r[1]} max={r[2]} avg={r[3]:.0f} vgpr={r[5]} agpr={r[6]} sgpr={r[7]} lds={r[8]} grid={r[9]} wg={r[10]}")
except Exception as e:  # schema differences between rocprof versions
    print("# (no per-dispatch table:", e, ")")
