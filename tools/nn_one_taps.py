#!/usr/bin/env python3
"""Shader-clock taps of rn_nn_one_kernel (instrumented build): where one stream's network spends its time.
usage: tools/nn_one_taps.py [streams = 1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rnnoise_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
capi.instrumented().__enter__()
m = capi.Model(bench.load_blob())
b = capi.Batch(m, n)
b.set_nn_path(0)
b.debug_pitch(arm_only=True)
pcm = synth.batch_pcm(range(n), 12)
for t in range(12):
    b.process(pcm[t:t + 1])
d = b.debug_pitch()
names = ["DMA conv1 + inputs", "conv1", "conv2", "L0 barrier", "L0 pack + barrier", "L0 row products", "L0 exchange barrier", "L0 gates"]
rows, chain = d[:, 1360:1368], d[:, 1368:1376]
print(f"rn_nn_one_kernel, {n} stream(s), shader clocks (100 MHz s_memtime ticks x 24 = 2.4 GHz cycles?) -- thread 0 | chain wave")
for k, nm in enumerate(names):
    print(f"  {nm:<34}{rows[:, k].mean():>10.0f}{chain[:, k].mean():>10.0f}")
print("  (chain wave, from slot 5 on: waited for its DMA | chain segment 0 | segment 2 requested + exchange barrier)")
print("  arrival of waves 0..13 at layer 0's exchange barrier (clocks after its pack barrier):", " ".join(f"{v:.0f}" for v in d[:, 1376:1390].mean(0)))
