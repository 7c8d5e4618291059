#!/usr/bin/env python3
"""Kernels and memory copies of a host-fed run on one time axis, from the CSV output of
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python bench.py --host-io [--s16] ...
What to look for: `__amd_rocclr_copyBuffer` (the runtime's blit kernel: a copy that found the DMA engine busy), the duration
of rn_analysis_kernel beside it, which queue each kernel ran on, and the period between consecutive synthesis kernels.

usage: tools/copy_timeline.py <kernel_trace.csv> <memory_copy_trace.csv> [rows = 60] [start fraction = 0.6]"""
import collections
import csv
import sys

kn = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("rn_") or "copyBuffer" in r["Kernel_Name"]]
cp = list(csv.DictReader(open(sys.argv[2])))
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.6
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
by = collections.defaultdict(list)
for r in kn:
    by[r["Kernel_Name"][:28] + " q" + r["Queue_Id"]].append(dur(r))
print("# kernels: calls, mean / min / max duration (us), by name and hardware queue")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:<36}{len(v):>5}{sum(v) / len(v):>10.1f}{min(v):>10.1f}{max(v):>10.1f}")
byc = collections.defaultdict(list)
for r in cp:
    if dur(r) > 100:
        byc[r["Direction"][12:]].append(dur(r))
print("# DMA copies longer than 100 us: calls, mean duration (us)")
for k, v in byc.items():
    print(f"  {k:<36}{len(v):>5}{sum(v) / len(v):>10.1f}")
syn = sorted(int(r["End_Timestamp"]) for r in kn if r["Kernel_Name"].startswith("rn_synthesis"))
gaps = sorted((b - a) / 1e3 for a, b in zip(syn, syn[1:]))
if gaps:
    print(f"# period between consecutive synthesis kernels: median {gaps[len(gaps) // 2]:.0f} us (min {gaps[0]:.0f})")
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"][12:]) for r in cp]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:26] + " q" + r["Queue_Id"]) for r in kn]
ev.sort()
i0 = int(len(ev) * frac)
t0 = ev[i0][0]
print(f"# timeline from event {i0} (us from there): start, end, duration, what   [events shorter than 20 us omitted]")
for a, b, n in ev[i0:i0 + rows]:
    if b - a >= 20000:
        print(f"{(a - t0) / 1e3:10.1f}{(b - t0) / 1e3:10.1f}{(b - a) / 1e3:9.1f}  {n}")
