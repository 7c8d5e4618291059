#!/usr/bin/env python3
"""Print the kernel timeline (start / end relative to a frame's first kernel) of a few steady-state frame steps out of a
rocprofv3 (rocpd sqlite) kernel trace of `bench.py`: which kernels of neighbouring frames actually overlap.

usage: tools/timeline.py <results.db> [n_kernels] [start fraction]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4   # where in the trace to start (bench.py: the last quarter is its one-stream pass)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
t0c, t1c = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = db.execute(f"select name, {t0c}, {t1c}, queue_id, stream_id from kernels where name like 'rn_%' order by {t0c}").fetchall() \
    if "stream_id" in cols else db.execute(f"select name, {t0c}, {t1c}, 0, 0 from kernels where name like 'rn_%' order by {t0c}").fetchall()
k0 = int(frac * len(rows))
base = rows[k0][1]
print(f"# {len(rows)} rn_ kernels in the trace; showing {n} from index {k0}; times in microseconds from the first one shown")
print(f"{'kernel':<28}{'start':>10}{'end':>10}{'dur':>9}  stream/queue")
for name, a, b, q, s in rows[k0:k0 + n]:
    print(f"{name:<28}{(a - base) / 1e3:>10.1f}{(b - base) / 1e3:>10.1f}{(b - a) / 1e3:>9.1f}  {s}/{q}")
