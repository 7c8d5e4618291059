#!/usr/bin/env python3
"""Register-resident FFT (rnnoise_amd/csrc/fft_reg.h) on the GPU: exchange-primitive map, bit parity with the oracle FFT,
shader clocks per transform for a lone wave and for a full machine, against the LDS work-area FFT (variant 2).

usage: tools/fft_bench.py            (needs a GPU)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle  # noqa: E402  (checker)
from rnnoise_amd import capi  # noqa: E402

L = capi.instrumented().__enter__()  # probe kernels: librnnoise_amd_instr.so
rng = np.random.Generator(np.random.PCG64(5))


def run(variant, x, reps, want_x=False):
    n = x.shape[0]
    out = np.empty_like(x)
    clk = np.zeros(n, np.uint64)
    xl = np.zeros((2, 6, 64), np.int32)
    t0 = time.perf_counter()
    rc = L.rnnoise_amd_debug_fft(0, variant, capi._fp(out), capi._fp(x), n, reps, clk.ctypes.data_as(C.POINTER(C.c_ulonglong)),
                                 xl.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0
    return out, clk, xl, time.perf_counter() - t0


x = np.zeros((8, 960, 2), np.float32)
x[1:5, :, 0] = rng.normal(0, 3000, (4, 960))
x[5:] = rng.normal(0, 1, (3, 960, 2))
x[7, ::3] = 0
want = np.stack([Oracle.fft(v.reshape(-1).copy()).reshape(960, 2) for v in x])
_, _, xl, _ = run(1, x, 1)
for var in (0, 1):
    for k, m in enumerate((1, 2, 4, 8, 16, 32)):
        ok = np.array_equal(xl[var, k], np.arange(64) ^ m)
        print(f"xlane variant {var} xor {m:2d}: {'ok' if ok else 'WRONG ' + str(xl[var, k].tolist())}")
for var in (0, 1, 2):
    got, _, _, _ = run(var, x, 1)
    ne = got.view(np.uint32) != want.view(np.uint32)
    print(f"variant {var}: {'bit-identical to the oracle FFT' if not ne.any() else str(int(ne.sum())) + ' words differ, first ' + str(np.argwhere(ne)[:3].tolist())}")
for n in (1, 65536):
    xs = np.ascontiguousarray(np.tile(x, ((n + 7) // 8, 1, 1))[:n])
    for var in (2, 0, 1, 14, 15, 16, 18):
        run(var, xs, 2)
        _, clk, _, wall = run(var, xs, 9)
        _, clk1, _, wall1 = run(var, xs, 1)
        per = (clk.astype(np.float64) - clk1) / 8
        print(f"n={n:6d} variant {var}: {per.mean():9.0f} clocks per transform per wave (min {per.min():.0f} max {per.max():.0f}); "
              f"wall delta {1e3 * (wall - wall1):.2f} ms for 8 x {n} transforms")
