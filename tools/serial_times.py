#!/usr/bin/env python3
"""Stand-alone kernel durations: the four kernels of a frame step run back to back on ONE stream
(single-frame calls switch the 3-stream pipeline off), HIP events around each of K0/K1/K2/K3.

usage: tools/serial_times.py [streams ...]     (needs a GPU; bench.py reports the pipelined numbers)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rnnoise_amd import capi  # noqa: E402

dev = torch.device("cuda:0")
model = capi.Model(bench.load_blob())
vector = "--vector" in sys.argv  # the vector-path network (rn_nn_one_kernel up to 256 streams) instead of the batch default
for N in [int(x) for x in sys.argv[1:] if not x.startswith("-")] or [4096, 65536]:
    b = capi.Batch(model, N)
    if vector:
        b.set_nn_path(0)
    cap = 8
    d_in = bench.synth_pcm_torch(torch, N, cap, dev, seed_base=0)
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty((cap, N), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    esz = N * 480 * 4

    def run(k0, n):
        for f in range(k0, k0 + n):
            k = f % cap
            b.process_device(d_out.data_ptr() + k * esz, d_in.data_ptr() + k * esz, d_vad.data_ptr() + k * N * 4, 0, 1, st)

    run(0, 6)
    torch.cuda.synchronize()
    b.enable_timing(True)
    K = 20
    t0 = time.perf_counter()
    run(6, K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    k = b.kernel_ms()
    rest = dt - k["analysis"] - k["network"] - k["synthesis"] - k["highpass"]
    print(f"N={N}: step {dt:.4f} ms serial | K0 {k['highpass']:.4f}  K1 {k['analysis']:.4f}  K2 {k['network']:.4f}  "
          f"K3 {k['synthesis']:.4f}  launch gaps {rest:.4f}")
    b.close()
