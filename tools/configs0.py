#!/usr/bin/env python3
"""BASELINE configs[0] through the drop-in rnnoise.h API on the GPU: one stream, 10 s (1000 frames), frame by frame.
Steady-state frames/s of the device-resident path (rnnoise_create) and of the self-contained caller-memory path
(rnnoise_get_size + rnnoise_init), one thread and four threads; the first 100 frames are warm-up."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rnnoise_amd import capi, synth  # noqa: E402

L = capi.lib()
L.rnnoise_process_frame.restype = C.c_float
model = capi.Model(bench.load_blob())
pcm = synth.stream_pcm(1, 1100).astype(np.float32).reshape(1100, 480)
FP = C.POINTER(C.c_float)


def run(handle, frames):
    buf = np.empty(480, np.float32)
    p = buf.ctypes.data_as(FP)
    for t in frames:
        buf[:] = pcm[t]
        L.rnnoise_process_frame(handle, p, p)


def make(kind):
    if kind == "pooled":
        return C.c_void_p(L.rnnoise_create(model.h)), None
    mem = (C.c_char * L.rnnoise_get_size())()
    assert L.rnnoise_init(C.cast(mem, C.c_void_p), model.h) == 0
    return C.cast(mem, C.c_void_p), mem


for kind in ("pooled", "caller-memory"):
    for threads in (1, 4):
        hs = [make(kind) for _ in range(threads)]
        for h, _ in hs:
            run(h, range(100))
        th = [threading.Thread(target=run, args=(h, range(100, 1100))) for h, _ in hs]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        print(f"configs[0] {kind:<14} {threads} thread(s) x 1000 frames: {dt:.3f} s = {threads * 1000 / dt:8.0f} frames/s "
              f"({1e6 * dt / 1000:.0f} us per frame per thread; real time needs 100 frames/s per stream)")
        if kind == "pooled":
            for h, _ in hs:
                L.rnnoise_destroy(h)
