#!/usr/bin/env python3
"""rn_analysis_kernel by section from the ASSEMBLY (no GPU needed): dsp_kernels.hip is compiled with -DRN_K1_MARKS=1, which
turns every K1_STOP(k) into a `; K1MARK k` comment, and the instructions between consecutive marks are counted -- VALU (with
the issue classes of tools/valu_mix.py), SALU, LDS and global-memory instructions.  Loops (a backward branch) are listed with
their body size; their trip counts come from LOOP_TRIPS below (the source's constant loop bounds), so the totals are an
estimate of the DYNAMIC per-wave counts that profiles/r*_k1_sections*.txt measure with PMC counters.

usage: tools/asm_sections.py [--kernel rn_analysis_kernel] [--flags "..."] [--keep file.s]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from valu_mix import COST, classify  # noqa: E402

NAMES = ["window X", "FFT X", "store X + Ex", "downsample+FIR", "y4/Z + coarse pass 1", "coarse pass 2", "narrow 1 + coarse select",
         "shifted copy", "narrow 2+3", "fine select", "doubling prep", "doubling dots", "decide", "window P + loads",
         "FFT P", "store P + Ep + Exp", "features (rest)"]

# trip counts of the loops that survive unrolling, by section and order of appearance (source loop bounds; data-dependent
# loops get a typical count).  Sections not listed: every loop body counted once.
LOOP_TRIPS = {
    "store X + Ex": [2, 7],                  # band_sums: 8 slots per trip for the chains that long, then one slot per trip
    "y4/Z + coarse pass 1": [15],            # chain_dot8_x3: 240 / 8 steps, unrolled by 2
    "narrow 1 + coarse select": [3],         # fbp_sweep_row: 3 blocks of 64 steps (best_pitch_select's loops: counted once)
    "narrow 2+3": [30, 0, 10],               # chain_sq8 (16 steps per trip); the one-block-ahead chain (A/B only); chain_dot16_xrow<deep>: 10 x 48
                                             # steps.  sweep_syy_fine_row_x's 5 blocks close on a branch the rule below does not see: counted once
    "fine select": [4, 4, 4, 4, 4],
    "doubling dots": [6, 30, 0],             # sweep_yy_lookup_row_x: 6 blocks of 64 steps; chain_dot16_xrow 480 / 16; the per-lane-x variant is not taken
    "decide": [0],                           # (the second two-lane pass runs only when a shorter period wins)
    "store P + Ep + Exp": [2, 7],
}
# sections that one or two waves run for the K1_SPW = 4 streams of their workgroup: per-wave share of their loops
SHARED = {"narrow 1 + coarse select": [0.25], "narrow 2+3": [0.25, 0.25, 0.25], "doubling dots": [0.25, 0.5, 1]}


def compile_asm(flags, keep):
    out = keep or os.path.join(tempfile.mkdtemp(), "dsp_marks.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
           "-fno-slp-vectorize", "-DRN_K1_MARKS=1", "-S", "--cuda-device-only", "-o", out,
           os.path.join(ROOT, "rnnoise_amd", "csrc", "dsp_kernels.hip")] + flags.split()
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def kernel_lines(path, kernel):
    lines, on = [], False
    for ln in open(path):
        if ln.startswith(kernel + ":"):
            on = True
            continue
        if on:
            if ln.startswith(".Lfunc_end"):
                break
            lines.append(ln.rstrip("\n"))
    return lines


def kind_of(mn):
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def analyse(lines, with_trips=True):
    # split into sections at the marks
    # a section = the code in front of a mark, named after the mark's number
    sections, marks = [[]], []
    for ln in lines:
        m = re.match(r"\s*; K1MARK (\d+)", ln)
        if m:
            marks.append(int(m.group(1)))
            sections.append([])
            continue
        sections[-1].append(ln)
    rows = []
    for si, sec in enumerate(sections):
        mk = marks[si] if si < len(marks) else None
        name = NAMES[mk - 1] if mk and mk - 1 < len(NAMES) else NAMES[-1]
        # instruction list with label positions
        labels, insts, headers = {}, [], set()
        for ln in sec:
            t = ln.strip()
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                labels[m.group(1)] = len(insts)
                if "Loop Header" in t:
                    headers.add(m.group(1))
                continue
            if not t or t.startswith((";", ".", "//")) or ":" in t.split()[0]:
                continue
            parts = t.split(None, 1)
            insts.append((parts[0], parts[1] if len(parts) > 1 else ""))
        weight = [1.0] * len(insts)
        ends = {}
        for i, (mn, ops) in enumerate(insts):
            if mn.startswith("s_cbranch") or mn == "s_branch":
                tgt = ops.strip().split()[0] if ops else ""
                if tgt in headers and tgt in labels and labels[tgt] <= i:
                    ends[tgt] = i   # the last backward branch to a loop header closes the loop
        loops = sorted((labels[t], e) for t, e in ends.items())
        trips = LOOP_TRIPS.get(name, []) if with_trips else []
        share = SHARED.get(name, [])
        for li, (a, b) in enumerate(loops):
            tr = (trips[li] if li < len(trips) else 1) * (share[li] if li < len(share) else 1)
            for k in range(a, b + 1):
                weight[k] *= tr
        tot = {"valu": 0.0, "salu": 0.0, "lds": 0.0, "vmem": 0.0, "other": 0.0}
        cls = {"fast": 0.0, "std": 0.0, "trans": 0.0}
        for (mn, ops), w in zip(insts, weight):
            k = kind_of(mn)
            tot[k] += w
            if k == "valu":
                cls[classify(mn, ops)] += w
        clk = sum(cls[c] * COST[c] for c in cls)
        rows.append((name, tot, cls, clk, [(b - a + 1, (trips[li] if li < len(trips) else 1)) for li, (a, b) in enumerate(loops)]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="rn_analysis_kernel")
    ap.add_argument("--flags", default="")
    ap.add_argument("--keep", default="")
    ap.add_argument("--asm", default="", help="an already compiled .s")
    a = ap.parse_args()
    path = a.asm or compile_asm(a.flags, a.keep)
    rows = analyse(kernel_lines(path, a.kernel))
    print(f"# {a.kernel}: per-wave instruction estimate by section (static assembly x loop trip counts; tools/asm_sections.py)")
    print(f"{'section':<28}{'VALU':>7}{'fast':>7}{'std':>7}{'trans':>6}{'clk':>8}{'SALU':>7}{'LDS':>6}{'VMEM':>6}  loops (body x trips)")
    T = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "clk": 0}
    for name, tot, cls, clk, loops in rows:
        print(f"{name:<28}{tot['valu']:>7.0f}{cls['fast']:>7.0f}{cls['std']:>7.0f}{cls['trans']:>6.0f}{clk:>8.0f}{tot['salu']:>7.0f}{tot['lds']:>6.0f}"
              f"{tot['vmem']:>6.0f}  {' '.join(f'{b}x{t}' for b, t in loops)}")
        for k in ("valu", "salu", "lds", "vmem"):
            T[k] += tot[k]
        T["clk"] += clk
    print(f"{'whole kernel':<28}{T['valu']:>7.0f}{'':>20}{T['clk']:>8.0f}{T['salu']:>7.0f}{T['lds']:>6.0f}{T['vmem']:>6.0f}"
          f"   mean {T['clk'] / max(1, T['valu']):.2f} clk / VALU")


if __name__ == "__main__":
    main()
