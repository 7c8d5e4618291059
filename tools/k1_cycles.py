#!/usr/bin/env python3
"""Print the shader-clock breakdown of rn_analysis_kernel (debug taps) for a lone wave and under load."""
import lzma, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnnoise_amd import capi, synth
NAMES = ["load", "-", "win+FFT(X)+Ex", "downsample+FIR", "coarse xcorr", "coarse scan", "fine xcorr",
         "fine select", "doubling dots", "Syy+yy sweeps", "decide", "P FFT+Ep+Exp+feat"]
blob = lzma.decompress(open(os.path.join(ROOT, "tests/golden/default.blob.xz"), "rb").read())
capi.instrumented().__enter__()  # the taps are compiled into librnnoise_amd_instr.so only
m = capi.Model(blob)
for n in (1, int(sys.argv[1]) if len(sys.argv) > 1 else 4096):
    b = capi.Batch(m, n)
    b.debug_pitch(arm_only=True)
    b.set_nn_path(2 if "--layers" in sys.argv and n >= 64 else 1)
    pcm = np.ascontiguousarray(np.tile(synth.batch_pcm(range(min(n, 16)), 6), (1, (n + 15) // 16, 1))[:, :n])
    b.process(pcm)
    d = b.debug_pitch()
    if "--nn" in sys.argv:
        clk2 = d[::16, 1360:1367]
        print(f"--- N={n}: MFMA network kernel, mean shader clocks per phase over tiles (total {clk2.sum(1).mean():.0f}) ---")
        for k, name in enumerate(["load+quantise", "conv1", "conv2", "gru1", "gru2", "gru3", "dense/vad"]):
            print(f"  {name:<20} {clk2[:, k].mean():>10.0f}  {100 * clk2[:, k].mean() / clk2.sum(1).mean():5.1f}%")
        if n >= 64 and "--layers" in sys.argv:
            c = d[::64, 1367:1376]
            print("    layer-wise GRU kernel (wave 0 of each 64-stream workgroup): prologue issue | wait to barrier | 3 unit tiles")
            for k in range(3):
                print(f"    gru{k+1}: {c[:, 3*k].mean():9.0f} {c[:, 3*k+1].mean():9.0f} {c[:, 3*k+2].mean():9.0f}")
            u = d[::64, 1376:1391]
            print("    gru1, inside a unit tile (wave 0): input gates | conversion | recurrent gates | wait + rows + conversion | activations + stores")
            for ui in range(3):
                print(f"    unit tile {ui}: " + " ".join(f"{u[:, 5 * ui + i].mean():9.0f}" for i in range(5)))
            x = d[::64, 1391:1395]
            if x.any():  # (the fold form of the layer kernel: the output chains' burst behind each unit tile, and the tail)
                print("    output chains (wave 0): burst behind unit tile 0 | 1 | 2 | tail behind the last unit tile")
                print("    " + " ".join(f"{x[:, i].mean():9.0f}" for i in range(4)))
    clk = d[:, 1348:1360]
    print(f"--- N={n}: mean shader clocks per section over streams (total {clk.sum(1).mean():.0f}) ---")
    for k, name in enumerate(NAMES):
        print(f"  {name:<20} {clk[:, k].mean():>10.0f}  {100 * clk[:, k].mean() / clk.sum(1).mean():5.1f}%")
    b.close()
