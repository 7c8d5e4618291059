#!/usr/bin/env python3
"""VERDICT r4 Next #4, measured: how much of a frame step's kernel time runs UNDER other kernels' time.
For each configuration (environment switches of the library) `bench.py` is run once at 65,536 streams; from its line:
  step          ms per frame step in the pipelined schedule (three streams: high-pass two frames ahead, analysis(f+1) beside network + synthesis(f))
  sum alone     K0 + K1 + K2 + K3 with every kernel on ONE stream (their stand-alone durations)
  hidden        sum alone - step: kernel time that ran under another kernel's
  inside        the kernels' durations INSIDE the pipeline (HIP events): what co-residency stretches them to
and, from the built code objects, what decides whether two kernels CAN share a CU (registers per wave, LDS per workgroup).

usage (GPU box): tools/overlap_table.py > gpurun_out/.../overlap.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = [
    ("default (K1 4 workgroups / CU, GRU w4)", {}),
    ("one stream (no overlap at all)", {"RNNOISE_AMD_PIPE": "9"}),
    ("only the high-pass aside", {"RNNOISE_AMD_PIPE": "1"}),
    ("K1 at 3 workgroups / CU (52 KB each: 46 KB of LDS and 176 VGPRs per SIMD left for K0 / K3 / front waves)", {"RNNOISE_AMD_K1_LDS": "13000"}),
    ("K1 at 2 workgroups / CU", {"RNNOISE_AMD_K1_LDS": "20000"}),
    ("GRU w8 (152 KB: a layer workgroup owns its CU)", {"RNNOISE_AMD_GRU_VARIANT": "w8"}),
    ("GRU v3 (12 waves, 145 VGPRs, persistent)", {"RNNOISE_AMD_GRU_VARIANT": "v3"}),
    ("K1 at 3 workgroups / CU + GRU w8", {"RNNOISE_AMD_K1_LDS": "13000", "RNNOISE_AMD_GRU_VARIANT": "w8"}),
]


def bench(env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-parity", "--steps", "40", "--warmup", "8", "--repeats", "9"],
                       # (RNNOISE_AMD_K1_LDS and the lab forms of the layer kernel exist in the instrumented library only: every row is measured on it)
                       env=dict(os.environ, RNNOISE_AMD_LIB=os.path.join(ROOT, "rnnoise_amd", "librnnoise_amd_instr.so"), **env), capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def main():
    print("# tools/overlap_table.py: bench.py --steps 40 --repeats 9 at 65,536 streams per configuration, one box, one call")
    print(f"# {'configuration':<100s} | M frames/s | step ms | sum alone | hidden | alone: K0 K1 K2 K3 | inside the pipeline: K0 K1 K2 K3")
    for name, env in CONFIGS:
        d = bench(env)
        if not d:
            print(f"{name:<102s} | FAILED")
            continue
        alone = d.get("roofline_standalone", {}).get("kernel_ms")
        inside = d["roofline"]["kernel_ms"]
        order = ("highpass", "analysis", "network", "synthesis")
        fmt = lambda k: " ".join(f"{k[x]:.3f}" for x in order) if k else "-"
        s = sum(alone[x] for x in order) if alone else float("nan")
        print(f"{name:<102s} | {d['value'] / 1e6:10.2f} | {d['ms_per_step']:7.4f} | {s:9.4f} | {s - d['ms_per_step']:6.3f} | {fmt(alone)} | {fmt(inside)}", flush=True)
    # what decides co-residency (needs the object files: they do not travel to the GPU box -- run this part where the library was built)
    try:
        import test_kernel_budgets_cpu as t
        rows = []
        facts = [("dsp_kernels", "rn_analysis_kernel", 4, 38.0), ("hp_kernel", "rn_hp_kernel", 1, 0.0), ("dsp_kernels", "rn_synthesis_kernel", 1, 4.9),
                 ("nn_mfma", "rn_nn_front_kernel", 8, 33.0), ("nn_layers", "rn_nn_gru_kernel", 4, 72.0), ("nn_layers", "rn_nn_gru_w8_kernel", 8, 152.0),
                 ("nn_layers", "rn_nn_gru3_kernel", 12, 152.0), ("nn_layers", "rn_nn_dense_kernel", 8, 78.0)]
        for obj, k, w, lds in facts:
            meta, _ = t._kernels(os.path.join(t.BUILD, obj + ".o"))
            v = meta[k]["vgpr_count"]
            alloc = (v + 7) // 8 * 8
            per_simd = 512 // alloc
            wg = min(per_simd * 4 // w, int(160 // lds) if lds else 99)
            waves = wg * w // 4 if w >= 4 else min(per_simd, 8)
            rows.append(f"# {k:<28s} {w:8d}  {v:10d}  {lds:11.1f}  -> {wg} workgroup(s) = {waves} wave(s) per SIMD, {waves * alloc} of 512 VGPRs, {wg * lds:.0f} of 160 KB")
        print("#\n# co-residency facts from the code objects (a CU: 4 SIMDs x 512 VGPRs per lane, 160 KB of LDS):")
        print("# kernel                       waves/wg  VGPRs/wave  LDS/wg (KB)  -> alone on a CU")
        print("\n".join(rows))
    except Exception as e:  # noqa: BLE001
        print(f"# (co-residency facts: object files not present here: {type(e).__name__})")


if __name__ == "__main__":
    main()
