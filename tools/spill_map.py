#!/usr/bin/env python3
"""Where a kernel's scratch (spill) instructions sit relative to its MFMA blocks, barriers and branches (no GPU):
tools/spill_map.py <object> <kernel>"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_kernel_budgets_cpu as t
with tempfile.TemporaryDirectory() as td:
    co = t._code_object(os.path.join(os.environ.get("KMETA_BUILD", t.BUILD), sys.argv[1] + ".o"), td)
    dis = subprocess.run([f"{t.LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
cur, lines = None, []
for ln in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\w+)>:", ln)
    if m:
        cur = m.group(1)
        continue
    if cur == sys.argv[2] and "\t" in ln:
        lines.append(ln.split("\t")[1] if len(ln.split("\t")) > 1 else "")
out = []
for i, l in enumerate(lines):
    op = l.split()[0] if l.split() else ""
    tag = None
    if op.startswith("scratch_store"): tag = "S"
    elif op.startswith("scratch_load"): tag = "L"
    elif op.startswith("v_mfma"): tag = "M"
    elif op.startswith("s_barrier"): tag = "|BARRIER|"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): tag = "<br>"
    elif op.startswith("global_load_lds"): tag = "d"
    elif op.startswith("global_store"): tag = "w"
    elif op.startswith("s_setprio"): tag = "<prio>"
    if tag:
        if out and out[-1][1] == tag and len(tag) == 1:
            out[-1][2] += 1
        else:
            out.append([i, tag, 1])
print(f"{sys.argv[2]}: {len(lines)} instructions; S = scratch store, L = scratch load, M = MFMA, d = LDS-DMA, w = global store")
print(" ".join(f"{tag}{n if n > 1 else ''}@{i}" if len(tag) == 1 else f"{tag}@{i}" for i, tag, n in out))
