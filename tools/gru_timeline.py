#!/usr/bin/env python3
"""Timeline of ONE GRU-layer workgroup, every wave (instrumented build, gru_body2 variants; $RNNOISE_AMD_GRU_TIMELINE=1): shader clocks,
relative to the workgroup's first wave, at kernel entry, behind the prologue's barrier, and at the six boundaries of each unit tile
(start | input gates | conversion | recurrent gates | wait + rows + conversion | activations + stores).  Waves w and w + 4 share a SIMD.

usage: RNNOISE_AMD_GRU_VARIANT=<o0|bd|...> tools/gru_timeline.py [streams] [block ...]"""
import lzma
import os
import sys

import numpy as np

os.environ["RNNOISE_AMD_GRU_TIMELINE"] = "1"
os.environ.setdefault("RNNOISE_AMD_GRU_VARIANT", "o0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnnoise_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
blocks = [int(x) for x in sys.argv[2:]] or [0, n // 64 // 2, n // 64 - 1]
blob = lzma.decompress(open(os.path.join(ROOT, "tests/golden/default.blob.xz"), "rb").read())
capi.instrumented().__enter__()
m = capi.Model(blob)
b = capi.Batch(m, n)
b.debug_pitch(arm_only=True)
b.set_nn_path(2)
b.set_schedule(9)  # one stream: the layer kernel alone on the machine
pcm = np.ascontiguousarray(np.tile(synth.batch_pcm(range(16), 4), (1, (n + 15) // 16, 1))[:, :n])
b.process(pcm)
d = b.debug_pitch().view(np.uint32)
print(f"# variant {os.environ['RNNOISE_AMD_GRU_VARIANT']}, {n} streams, one stream schedule; clocks relative to the workgroup's earliest wave")
V3 = os.environ["RNNOISE_AMD_GRU_VARIANT"].startswith("v3")
if V3:  # gru_body3: 12 waves, 2 unit tiles of 8 boundaries, entry / barrier in words 38 / 39, the second group + 40
    names = ["start", "zr-in", "conv", "zr-rec", "rows+sigm", "c-in", "c-rec", "tanh+store"]
    NW, UT, NB, STRIDE, E0, G2 = 12, 2, 8, 10, 38, 40
else:
    names = ["start", "in-gates", "conv", "rec-gates", "rows+conv", "act+store"]
    NW, UT, NB, STRIDE, E0, G2 = 8, 3, 6, 6, 18, 20
for blk in blocks:
    rows = d[blk * 64 + 1: blk * 64 + 1 + NW, :80].astype(np.int64)
    t0 = rows[:, E0].min()
    rel = (rows - t0) & 0xffffffff
    print(f"workgroup {blk}:")
    print("  wave  entry barrier | " + " | ".join(f"unit tile {u}: " + " ".join(f"{x:>9s}" for x in names) for u in range(UT)))
    for w in range(NW):
        print(f"  {w:4d} {rel[w, E0]:6d} {rel[w, E0 + 1]:7d} | " + " | ".join(" ".join(f"{rel[w, STRIDE * u + i]:9d}" for i in range(NB)).rjust(13 + 10 * NB) for u in range(UT)))
    if rows[:, G2:G2 + STRIDE * UT].any():  # a persistent variant: the workgroup's second group
        print("  second group of the same workgroup:")
        for w in range(NW):
            print(f"  {w:4d}                | " + " | ".join(" ".join(f"{rel[w, G2 + STRIDE * u + i]:9d}" for i in range(NB)).rjust(13 + 10 * NB) for u in range(UT)))
    g0 = np.stack([rel[:, STRIDE * u: STRIDE * u + NB] for u in range(UT)], axis=1)
    ph = np.diff(g0, axis=2)
    print(f"  first group done at {g0[:, -1, -1].max()}; mean phase lengths over waves and unit tiles: " + ", ".join(f"{nm} {ph[:, :, i].mean():.0f}" for i, nm in enumerate(names[1:])))
    print("  per wave (mean over unit tiles): " + " ; ".join(f"w{w}: " + "/".join(f"{ph[w, :, i].mean():.0f}" for i in range(NB - 1)) for w in range(NW)))
b.close()
