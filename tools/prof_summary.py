#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the text table kept under profiles/.

usage: tools/prof_summary.py <results.db> [title]   (prints to stdout)
Columns mirror `rocprofv3 --kernel-trace --stats`: calls, total/avg/min/max duration (us), share.
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
    "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
    "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
print(f"# {title}")
print(f"# source: rocprofv3 --kernel-trace --stats (rocpd db), durations in microseconds")
print(f"{'kernel':<44} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6} "
      f"{'vgpr':>5} {'sgpr':>5} {'lds':>7} {'scr':>4} {'grid':>8} {'wg':>4}")
for n, c, tot, avg, mn, mx, vg, sg, lds, scr, gx, wx in rows:
    short = n if len(n) <= 44 else n[:41] + "..."
    print(f"{short:<44} {c:>6} {tot/1e3:>12.1f} {avg/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {100*tot/total:>6.2f} "
          f"{vg or 0:>5} {sg or 0:>5} {lds or 0:>7} {scr or 0:>4} {gx or 0:>8} {wx or 0:>4}")
