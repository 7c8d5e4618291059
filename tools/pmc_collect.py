#!/usr/bin/env python3
"""Collect hardware counters per kernel with rocprofv3, one pass per counter group (on the GPU box).

usage: tools/pmc_collect.py OUTDIR "CNT_A CNT_B,CNT_C ..." -- <command ...>
  groups are separated by ',', counters inside a group by spaces.  Each group is its own run of
  <command> under `rocprofv3 --kernel-trace --pmc ... --output-format csv` (never combined with other
  trace domains).  Prints mean counter value per kernel launch.
"""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

out, groups = sys.argv[1], [g.split() for g in sys.argv[2].split(",")]
cmd = sys.argv[sys.argv.index("--") + 1:]
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
acc = defaultdict(lambda: defaultdict(list))
for gi, g in enumerate(groups):
    d = os.path.join(out, f"pass{gi}")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *g, "--output-format", "csv", "-d", d, "--"] + cmd,
                       env=env, capture_output=True, text=True)
    if r.returncode:
        print(f"# pass {gi} ({' '.join(g)}) failed rc={r.returncode}: {r.stderr[-400:]}")
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            # duration of the same dispatch (this pass): GRBM_GUI_ACTIVE / duration = the shader clock the kernel ran at
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                acc[row["Kernel_Name"]]["GRBM_PASS_DURATION_NS"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    if "GRBM_GUI_ACTIVE" in g and not any("GRBM_PASS_DURATION_NS" in v for v in acc.values()):
        # older csv layout: durations from the kernel trace of the same pass, matched by dispatch order per kernel
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                acc[row["Kernel_Name"]]["GRBM_PASS_DURATION_NS"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
names = sorted({c for k in acc.values() for c in k})
print("kernel," + ",".join(names) + ",launches")
for k, cs in sorted(acc.items()):
    if not k.startswith("rn_"):
        continue
    n = max(len(v) for v in cs.values())
    print(k + "," + ",".join(f"{sum(cs[c]) / len(cs[c]):.1f}" if cs.get(c) else "" for c in names) + f",{n}")
