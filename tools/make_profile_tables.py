#!/usr/bin/env python3
"""Turn the raw output of tools/collect_profiles.sh (gpurun_out/prof/) into the files kept under profiles/.

usage: tools/make_profile_tables.py [gpurun_out/prof] [profiles] [prefix]
"""
import csv
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
pre = sys.argv[3] if len(sys.argv) > 3 else "r1_final"

for n in ("bench_4096", "bench_16384", "bench_32768", "bench_65536", "bench_4096_vector"):
    shutil.copy(os.path.join(src, n + ".json"), os.path.join(dst, f"{pre}_{n}.json"))
shutil.copy(os.path.join(src, "serial_times.txt"), os.path.join(dst, f"{pre}_serial_times.txt"))

# kernel stats: our kernels, everything else (bench.py's torch input synthesis) folded into one line
lines = open(os.path.join(src, "kernel_stats.txt")).read().split("\n")
out, other_calls, other_us = [], 0, 0.0
for l in lines:
    if l.startswith("#") or l.startswith("kernel ") or l.startswith("rn_"):
        out.append(l)
    elif l.strip():
        f = l.split()
        k = next(i for i, t in enumerate(f) if t.isdigit())
        other_calls += int(f[k])
        other_us += float(f[k + 1])
out.append(f"(torch kernels of bench.py's input synthesis, outside the timed region)  calls {other_calls}  total_us {other_us:.1f}")
open(os.path.join(dst, f"{pre}_kernel_stats.txt"), "w").write("\n".join(out) + "\n")

traffic = {}
for n_streams in (4096, 65536):
    rows = list(csv.DictReader(l for l in open(os.path.join(src, f"pmc_{n_streams}.csv")) if not l.startswith("#")))
    hdr = (f"# PMC counters, mean per kernel launch: rocprofv3 --kernel-trace --pmc <group> -- python bench.py --no-cpu-baseline "
           f"--streams {n_streams} --steps 6 --warmup 2  (MFMA path, pipelined)\n"
           "# one run per counter group (tools/pmc_collect.py); SQ_* are summed over all shader engines\n"
           "# HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is in KiB and on gfx950 reports 1/2 of wide coalesced reads "
           "(MI355X_MICROARCH.md, HBM section)\n")
    t = (f"{'kernel':<22}{'VALU/wave':>10}{'SALU/wave':>10}{'LDS/wave':>9}{'MFMA/wave':>10}{'VMEMrd/wave':>12}{'valu_busy%':>11}"
         f"{'wait%':>7}{'L1acc/launch':>14}{'L2hit%':>8}{'FETCH_KiB':>11}{'WRITE_KiB':>11}{'HBM_B/frame':>12}\n")
    for r in rows:
        def g(k):
            return float(r[k]) if r.get(k) else 0.0
        w = g("SQ_WAVES") or 1
        hbm = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / n_streams
        t += (f"{r['kernel']:<22}{g('SQ_INSTS_VALU') / w:>10.0f}{g('SQ_INSTS_SALU') / w:>10.0f}{g('SQ_INSTS_LDS') / w:>9.0f}"
              f"{g('SQ_INSTS_MFMA') / w:>10.0f}{g('SQ_INSTS_VMEM_RD') / w:>12.0f}"
              f"{100 * g('SQ_ACTIVE_INST_VALU') / (g('SQ_WAVE_CYCLES') or 1):>11.1f}{100 * g('SQ_WAIT_ANY') / (g('SQ_WAVE_CYCLES') or 1):>7.1f}"
              f"{g('TCP_TOTAL_CACHE_ACCESSES_sum'):>14.0f}{100 * g('TCC_HIT_sum') / ((g('TCC_HIT_sum') + g('TCC_MISS_sum')) or 1):>8.1f}"
              f"{g('FETCH_SIZE'):>11.0f}{g('WRITE_SIZE'):>11.0f}{hbm:>12.0f}\n")
        traffic.setdefault(str(n_streams), {})[r["kernel"].replace("_lean", "")] = {
            "hbm_bytes_per_frame": round(hbm, 1), "fetch_kib_per_launch": g("FETCH_SIZE"),
            "write_kib_per_launch": g("WRITE_SIZE"), "kernel": r["kernel"]}
    open(os.path.join(dst, f"{pre}_pmc_{n_streams}.txt"), "w").write(hdr + t)
json.dump({"source": f"profiles/{pre}_pmc_4096.txt, {pre}_pmc_65536.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                     "gfx950 x2 read correction); bench.py uses the set measured nearest to its batch size",
           "by_streams": traffic},
          open(os.path.join(dst, "r1_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{pre}_pmc_4096.txt")).read())
print(open(os.path.join(dst, f"{pre}_pmc_65536.txt")).read())
