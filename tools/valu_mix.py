#!/usr/bin/env python3
"""Mean issue cost of a kernel's VALU instructions, from its disassembly and the measured per-instruction table.

profiles/r4_valu_issue.txt (rnnoise_amd/csrc/tools/valu_issue.hip, run on the MI355X) shows that a wave64 VALU instruction
does NOT have one cost on gfx950: with two or more waves on a SIMD the plain f32 / u32 VOP2 forms issue every ~2.2-2.3 clocks,
anything with a DPP / SDWA modifier, an SGPR source, a packed, f64, integer-multiply, compare, select, convert or 3-operand
form every ~4.1, transcendentals and v_permlane32_swap every ~8.1.  The PMC ratio 4*SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU is
4.0 for every kernel (the counter ticks once per instruction per quad-cycle) and says nothing about this.

This tool classifies every VALU instruction in the gfx950 code object of the given objects by that table and prints, per
kernel, the static instruction mix and its mean cost.  The kernels' hot parts are straight-line (fully unrolled chains), so
the static mix is taken as the dynamic one: bound_cycles_per_wave = SQ_INSTS_VALU per wave (PMC) x mean cost.

usage: tools/valu_mix.py [--json profiles/valu_mix.json] [--k1-sections rnnoise_amd/csrc/build_instr/dsp_kernels.o k1_prefix.csv]
                         rnnoise_amd/csrc/build/*.o
  --k1-sections: rn_analysis_kernel's mean cost becomes a DYNAMIC one -- its sections priced separately and weighted by the
  per-section instruction counts of the PMC passes (the kernel's loops execute a mix that is not the file's average)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
# clocks per wave64 instruction per SIMD at >= 2 waves/SIMD ("B" column, 4 waves/SIMD) of profiles/r4_valu_issue.txt
COST = {"fast": 2.26, "std": 4.15, "trans": 8.12}
FAST = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_mul_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_xor_b32",
        "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32"}
MEASURED_FAST = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_sub_f32", "v_mov_b32", "v_and_b32", "v_xor_b32", "v_add_u32"}
TRANS = ("v_rcp_", "v_exp_", "v_log_", "v_sqrt_", "v_rsq_", "v_sin_", "v_cos_", "v_permlane32_swap")


def classify(mn: str, ops: str):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    if base.startswith(TRANS):
        return "trans"
    if mn.endswith(("_dpp", "_sdwa")):
        return "std"
    if base in FAST:
        srcs = ops.split(",")[1:]
        if any(re.match(r"\s*(s\d+|s\[|vcc|exec|ttmp|m0)", o) for o in srcs):
            return "std"   # an SGPR source costs the slow rate (row "v_add_f32 (sgpr operand)")
        return "fast"
    return "std"


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "dev.co"), os.path.join(td, "fat.bin")
        src = obj
        if not obj.endswith(".co"):
            subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], capture_output=True)
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if r.returncode or not os.path.exists(co) or not os.path.getsize(co):
                return {}   # no device code in this object (the host-side shim)
            src = co
        asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", src], check=True, capture_output=True, text=True).stdout
    cur, out = None, {}
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        f = line.strip().split(None, 1)
        if cur and f and re.match(r"^(v_|ds_|s_|global_|buffer_|flat_|scratch_)", f[0]):
            out[cur].append((f[0], f[1].split("//")[0] if len(f) > 1 else ""))
    return out


def k1_dynamic(instr_obj, csv_path):
    """Mean issue cost of rn_analysis_kernel weighted by what actually EXECUTES: the instrumented build's code is cut at its
    K1_STOP checks (s_cmp_eq_u32 <reg>, k -- the same boundaries tools/k1_prefix.sh measures between), every section's static
    mix is priced, and the sections are summed with the per-wave VALU counts the PMC passes MEASURED for them (k1_prefix.csv)
    as weights -- a section whose loops run 30 times counts 30 times, with the mix of its own code, not the file's average."""
    import csv
    ins = kernels_of(instr_obj).get("rn_analysis_kernel")
    if not ins:
        return None
    rows = list(csv.DictReader(open(csv_path)))
    cum = {int(r["stop"]): float(r["valu_per_wave"]) for r in rows}
    whole = cum.pop(0)
    stops = sorted(cum)
    dyn = [cum[k] - (cum[k - 1] if k > 1 else 0.0) for k in stops] + [whole - cum[stops[-1]]]
    # section boundaries: first `s_cmp_eq_u32 sN, k` for k = 1, 2, ... in program order
    cuts, want = [], 1
    for i, (mn, ops) in enumerate(ins):
        if mn == "s_cmp_eq_u32" and re.match(rf"\s*s\d+,\s*{want}\s*$", ops):
            cuts.append(i)
            want += 1
    if len(cuts) != len(stops):
        return None
    bounds = [0] + cuts + [len(ins)]
    tot_clk = tot_valu = 0.0
    per_section = []
    for si in range(len(bounds) - 1):
        sec = ins[bounds[si]:bounds[si + 1]]
        base = bounds[si]
        # loop membership by backward branches (objdump prints targets as <kernel+0xOFF>; fall back: no loops)
        inloop = [False] * len(sec)
        n = {"fast": 0.0, "std": 0.0, "trans": 0.0}
        nl = {"fast": 0.0, "std": 0.0, "trans": 0.0}
        for j, (mn, ops) in enumerate(sec):
            if mn.startswith("v_") and not mn.startswith(("v_mfma", "v_smfmac")):
                (nl if inloop[j] else n)[classify(mn, ops)] += 1
        stat = sum(n.values())
        d = max(dyn[si], 0.0)
        scale = d / stat if stat else 0.0  # (without loop information: the whole section scaled to its measured count)
        clk = sum(n[c] * COST[c] for c in n) * scale
        per_section.append((d, clk / d if d else 0.0))
        tot_clk += clk
        tot_valu += d
    return {"mean_cycles": round(tot_clk / tot_valu, 3) if tot_valu else None, "valu_dynamic_per_wave": round(tot_valu),
            "sections": [{"valu": round(d), "mean_cycles": round(c, 3)} for d, c in per_section]}


def main():
    args = sys.argv[1:]
    jpath = None
    k1 = None
    if args and args[0] == "--json":
        jpath, args = args[1], args[2:]
    if args and args[0] == "--k1-sections":  # --k1-sections <instrumented dsp_kernels.o> <k1_prefix.csv>
        k1, args = (args[1], args[2]), args[3:]
    res = {}
    for obj in args:
        for k, ins in kernels_of(obj).items():
            if not k.startswith("rn_"):
                continue
            n = {"fast": 0, "std": 0, "trans": 0}
            mfma = sel_vcc = unmeasured_fast = 0
            for mn, ops in ins:
                if not mn.startswith("v_"):
                    continue
                if mn.startswith(("v_mfma", "v_smfmac")):
                    mfma += 1
                    continue
                c = classify(mn, ops)
                n[c] += 1
                if mn == "v_cndmask_b32_e32":
                    sel_vcc += 1
                if c == "fast" and re.sub(r"_(e32|e64)$", "", mn) not in MEASURED_FAST:
                    unmeasured_fast += 1
            tot = sum(n.values())
            if not tot:
                continue
            mean = sum(n[c] * COST[c] for c in n) / tot
            res[k] = {"valu_static": tot, "fast": n["fast"], "std": n["std"], "trans": n["trans"], "mfma_static": mfma,
                      "select_on_vcc": sel_vcc, "fast_by_analogy": unmeasured_fast, "mean_cycles": round(mean, 3)}
    for r in res.values():
        r["weighting"] = "static"
    if k1 and "rn_analysis_kernel" in res:
        d = k1_dynamic(*k1)
        if d and d["mean_cycles"]:
            r = res["rn_analysis_kernel"]
            r["mean_cycles_static"] = r["mean_cycles"]
            r["mean_cycles"] = d["mean_cycles"]
            r["weighting"] = ("dynamic: the instrumented build's sections (K1_STOP boundaries) priced one by one and weighted by the VALU "
                              "instructions per wave the PMC passes measured for each (tools/k1_prefix.sh)")
            r["valu_dynamic_per_wave"] = d["valu_dynamic_per_wave"]
            r["sections"] = d["sections"]
    print(f"# mean VALU issue cost per kernel from the static instruction mix; classes: fast {COST['fast']} clk (plain f32/u32 VOP2, no SGPR "
          f"source), std {COST['std']} clk, trans {COST['trans']} clk  [profiles/r4_valu_issue.txt, >= 2 waves per SIMD]")
    print(f"{'kernel':<30}{'VALU':>7}{'fast':>7}{'std':>7}{'trans':>7}{'MFMA':>7}{'sel(vcc)':>9}{'mean clk':>10}")
    for k, r in sorted(res.items()):
        print(f"{k:<30}{r['valu_static']:>7}{r['fast']:>7}{r['std']:>7}{r['trans']:>7}{r['mfma_static']:>7}{r['select_on_vcc']:>9}{r['mean_cycles']:>10.2f}")
    if jpath:
        json.dump({"source": "tools/valu_mix.py over rnnoise_amd/csrc/build/*.o; costs from profiles/r4_valu_issue.txt", "cost": COST,
                   "kernels": res}, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
