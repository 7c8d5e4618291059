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
pre = sys.argv[3] if len(sys.argv) > 3 else "r3"

for n in ("bench_65536", "bench_4096", "bench_4096_vector", "bench_16384", "bench_little_32768", "bench_hostio_65536",
          "bench_hostio_s16_65536", "bench_s16_65536", "bench_4096_fpc1", "bench_16384_fpc1"):
    if os.path.exists(os.path.join(src, n + ".json")) and os.path.getsize(os.path.join(src, n + ".json")) > 10:
        shutil.copy(os.path.join(src, n + ".json"), os.path.join(dst, f"{pre}_{n}.json"))
for n in ("k1_sections.csv",):
    if os.path.exists(os.path.join(src, n)):
        shutil.copy(os.path.join(src, n), os.path.join(dst, f"{pre}_{n}"))
for n in ("serial_times", "section_taps_65536", "configs0", "configs0_cthreads", "k1_sections", "k1_narrow", "fft_bench", "network_schedules_65536", "pcie_peak",
          "valu_issue", "hostio_sdma", "hostio_breakdown", "k1_dots_conflicts", "gru_variants", "overlap"):
    if os.path.exists(os.path.join(src, n + ".txt")):
        shutil.copy(os.path.join(src, n + ".txt"), os.path.join(dst, f"{pre}_{n}.txt"))

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

by = {"default": {}, "little": {}}
for tag, model, n_streams, cmd in (("65536", "default", 65536, "--steps 4 --warmup 1 --repeats 2"),
                                   ("4096", "default", 4096, "--streams 4096 --steps 8 --warmup 2 --repeats 2"),
                                   ("little_32768", "little", 32768, "--model little --streams 32768 --steps 4 --warmup 1 --repeats 2")):
    f = os.path.join(src, f"pmc_{tag}.csv")
    if not os.path.exists(f):
        continue
    rows = list(csv.DictReader(l for l in open(f) if not l.startswith("#")))
    hdr = (f"# PMC counters, mean per kernel launch: RNNOISE_AMD_PIPE=9 rocprofv3 --kernel-trace --pmc <group> -- python bench.py --no-cpu-baseline {cmd}\n"
           "# (MFMA path, every kernel on one stream so that a kernel's counters are its own); one run per counter group (tools/pmc_collect.py);\n"
           "# SQ_* are summed over all shader engines, *_CYCLES in quad-cycles; cyc/VALU = 4*SQ_ACTIVE_INST_VALU/SQ_INSTS_VALU;\n"
           "# clock = GRBM_GUI_ACTIVE / XCDs / duration of the same pass (GHz); HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is in KiB\n"
           "# and on gfx950 reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section);\n"
           "# MFMA_busy% = SQ_VALU_MFMA_BUSY_CYCLES (cycles: 16 per v_mfma_i32_16x16x64_i8, 32 per v_mfma_f32_16x16x4_f32) / (1,024 SIMDs x the kernel's\n"
           "# GRBM_GUI_ACTIVE / 8 XCDs): the share of the launch during which a SIMD's matrix pipe is busy, averaged over the SIMDs\n")
    t = (f"{'kernel':<26}{'waves':>8}{'VALU/wave':>10}{'SALU/wave':>10}{'LDS/wave':>9}{'MFMA/wave':>10}{'VMEM/wave':>10}{'cyc/VALU':>9}"
         f"{'valu_act%':>10}{'wait%':>7}{'LDScyc/wave':>12}{'conflict%':>10}{'L2hit%':>8}{'FETCH_KiB':>11}{'WRITE_KiB':>11}{'HBM_B/frame':>12}{'clock_GHz':>10}{'MFMA_busy%':>11}\n")
    for r in rows:
        def g(k):
            return float(r[k]) if r.get(k) else 0.0
        w = g("SQ_WAVES") or 1
        hbm = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / n_streams
        cpi = 4 * g("SQ_ACTIVE_INST_VALU") / (g("SQ_INSTS_VALU") or 1)
        t += (f"{r['kernel']:<26}{w:>8.0f}{g('SQ_INSTS_VALU') / w:>10.0f}{g('SQ_INSTS_SALU') / w:>10.0f}{g('SQ_INSTS_LDS') / w:>9.0f}"
              f"{g('SQ_INSTS_MFMA') / w:>10.0f}{(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / w:>10.0f}{cpi:>9.2f}"
              f"{100 * g('SQ_ACTIVE_INST_VALU') / (g('SQ_WAVE_CYCLES') or 1):>10.1f}{100 * g('SQ_WAIT_ANY') / (g('SQ_WAVE_CYCLES') or 1):>7.1f}"
              f"{g('SQ_LDS_IDX_ACTIVE') / w:>12.0f}{100 * g('SQ_LDS_BANK_CONFLICT') / (g('SQ_LDS_IDX_ACTIVE') or 1):>10.1f}"
              f"{100 * g('TCC_HIT_sum') / ((g('TCC_HIT_sum') + g('TCC_MISS_sum')) or 1):>8.1f}"
              f"{g('FETCH_SIZE'):>11.0f}{g('WRITE_SIZE'):>11.0f}{hbm:>12.0f}")
        cl = next((g("GRBM_GUI_ACTIVE") / dv / g("GRBM_PASS_DURATION_NS") for dv in (1, 8)
                   if g("GRBM_PASS_DURATION_NS") and 0.5 <= g("GRBM_GUI_ACTIVE") / dv / g("GRBM_PASS_DURATION_NS") <= 3.0), 0.0)
        t += f"{cl:>10.3f}" if cl else f"{'':>10}"
        busy = 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE") / 8) if g("GRBM_GUI_ACTIVE") else 0.0
        t += f"{busy:>11.1f}\n" if g("SQ_INSTS_MFMA") else f"{'':>11}\n"
        # shader clock of the kernel: GRBM_GUI_ACTIVE (busy cycles, summed over the XCDs rocprofv3 reports) / its duration in
        # the same pass; the divisor (1 or 8 XCDs) is the one that lands in a shader clock's range
        clock = None
        if g("GRBM_GUI_ACTIVE") and g("GRBM_PASS_DURATION_NS"):
            for div in (1, 8):
                c = g("GRBM_GUI_ACTIVE") / div / g("GRBM_PASS_DURATION_NS")
                if 0.5 <= c <= 3.0:
                    clock = round(min(c, 2.4), 3)  # (the counter's window is a little wider than the kernel's: short kernels read high)
                    break
        by[model].setdefault(str(n_streams), {})[r["kernel"].replace("_single", "")] = {
            **({"clock_ghz": clock} if clock else {}),
            "hbm_bytes_per_frame": round(hbm, 1), "fetch_kib_per_launch": g("FETCH_SIZE"), "write_kib_per_launch": g("WRITE_SIZE"),
            "valu_per_wave": round(g("SQ_INSTS_VALU") / w), "valu_cycles_per_inst": round(cpi, 2),
            "lds_cycles_per_wave": round(g("SQ_LDS_IDX_ACTIVE") / w), "kernel": r["kernel"], "source": f"profiles/{pre}_pmc_{tag}.txt",
            **({"mfma_per_wave": round(g("SQ_INSTS_MFMA") / w), "mfma_busy_frac": round(busy / 100, 4)} if g("SQ_INSTS_MFMA") else {})}
    open(os.path.join(dst, f"{pre}_pmc_{tag}.txt"), "w").write(hdr + t)
    print(hdr + t)
json.dump({"source": f"profiles/{pre}_pmc_*.txt (rocprofv3 --pmc, separate passes, gfx950 x2 read correction); bench.py uses the set of its "
                     "model measured nearest to its batch size",
           "by_streams": by["default"], "little": by["little"]},
          open(os.path.join(dst, "pmc_by_streams.json"), "w"), indent=1)
