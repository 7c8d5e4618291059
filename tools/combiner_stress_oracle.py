#!/usr/bin/env python3
"""The seven checksums of tools/combiner_stress.sh from the ORACLE (CPU; the checker, not the product): the same generators, the same
1,000 frames per signal through oracle/rn_oracle.c on the rcpps profile named on the command line (the GPU box's CPUs are amd-zen5),
the same FNV-style checksum over every output sample and VAD value.  Equal to the GPU runs' = every one of those runs' states
produced the reference's bits for all 1,000 frames.

usage: tools/combiner_stress_oracle.py [profile = amd-zen5] [frames = 1000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
from oracle import binding  # noqa: E402

profile = sys.argv[1] if len(sys.argv) > 1 else "amd-zen5"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
binding.set_rcp_profile(profile)
blob = conftest.load_blob("default")
M32, M64 = 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF
sums = []
for k in range(7):
    g = (977 * k + 1) & M32
    x = np.empty((frames, 480), np.float32)
    for t in range(frames):
        for i in range(480):
            g = (g * 1664525 + 1013904223) & M32
            x[t, i] = float((g >> 18) - 8192)
    r = binding.Oracle(blob).run(x)
    h = 0xcbf29ce484222325
    out, vad = r["out"].view(np.uint32), r["vad"].view(np.uint32)
    for t in range(frames):
        for u in out[t].tolist():
            h = ((h ^ u) * 0x100000001b3) & M64
        h = ((h ^ int(vad[t])) * 0x100000001b3) & M64
    sums.append(h)
print(f"oracle ({profile}, {frames} frames per signal): checksums " + " ".join(f"{h:016x}" for h in sums))
