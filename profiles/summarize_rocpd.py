#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 (rocpd sqlite) result into a small text table.
usage: summarize_rocpd.py <results.db> [pattern]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else '%'
print(f"{'kernel':<40}{'calls':>8}{'total_us':>14}{'avg_us':>12}{'pct':>8}")
for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name[:39]:<40}{calls:>8}{tot:>14.1f}{avg:>12.1f}{pct:>8.2f}")
print()
print(f"{'kernel':<28}{'grid':>9}{'wg':>6}{'lds_B':>9}{'scratch_B':>10}{'vgpr':>6}{'agpr':>6}{'sgpr':>6}{'min_us':>10}{'avg_us':>10}{'max_us':>10}")
# one row per (kernel, grid): bench.py launches the BA kernels both on the 256-window batch and on a single window (its latency
# line), the two must not be averaged together
q = ("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, "
     "min(duration), avg(duration), max(duration), count(*) from kernels where name like ? group by name, grid_x order by name, grid_x")
for r in cur.execute(q, (pat,)):
    print(f"{r[0][:27]:<28}{r[1]:>9}{r[2]:>6}{r[3]:>9}{r[4]:>10}{r[5]:>6}{r[6]:>6}{r[7]:>6}{r[8]/1e3:>10.1f}{r[9]/1e3:>10.1f}{r[10]/1e3:>10.1f}   x{r[11]}")
