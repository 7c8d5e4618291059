#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / LDS of every kernel in a built object or library (code-object metadata).
usage: tools/kernel_resources.py rnnoise_amd/csrc/build/dsp_kernels.o [more objects...]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
for obj in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as td:
        out, fat = os.path.join(td, "dev.co"), os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], capture_output=True)
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(out) or not os.path.getsize(out):
            print(f"{obj}: no gfx950 code object ({r.stderr.strip()[:100]})")
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
        print(f"{os.path.basename(obj):<20} {g('name'):<34} vgpr {g('vgpr_count'):>4} agpr {blk.split()[0]:>3} sgpr {g('sgpr_count'):>4} "
              f"scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6} spill_v {g('vgpr_spill_count'):>3}")
