#!/usr/bin/env python3
"""A/B of the two MFMA network schedules on a GPU: bit comparison (fused tile kernel vs layer-wise) on ragged batches,
then stand-alone K2 time and pipelined throughput per schedule.   usage: tools/ab_layers.py [streams]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rnnoise_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
model = capi.Model(bench.load_blob())
SERIAL = "--serial" in sys.argv  # only the stand-alone loops (for rocprofv3 --kernel-trace: clean per-kernel averages)
if SERIAL:
    sys.argv.remove("--serial")
for n in () if SERIAL else (70, 16, 129):
    pcm = synth.batch_pcm(range(n), 12, lead_silence=2)
    res = []
    for path in (1, 2):
        b = capi.Batch(model, n)
        b.set_nn_path(path)
        res.append(b.process(pcm) + (b.export_state(n - 1),))
        b.close()
    ok = all(np.array_equal(np.asarray(a).view(np.uint32), np.asarray(c).view(np.uint32)) for a, c in zip(res[0][:3], res[1][:3]))
    ok = bool(ok) and bool(np.array_equal(np.asarray(res[0][3]), np.asarray(res[1][3])))
    print(f"n={n}: layer-wise == fused (pcm, vad, gains, state of the last stream): {ok}")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cap = 8
d_in = bench.synth_pcm_torch(torch, N, cap, dev, seed_base=0)
d_out = torch.empty_like(d_in)
d_vad = torch.empty((cap, N), device=dev)
st = torch.cuda.current_stream().cuda_stream
esz = N * 480 * 4
for path in (1, 2, 1, 2):
    b = capi.Batch(model, N)
    b.set_nn_path(path)
    for f in range(6):
        b.process_device(d_out.data_ptr() + f * esz, d_in.data_ptr() + f * esz, d_vad.data_ptr() + f * N * 4, 0, 1, st)
    torch.cuda.synchronize()
    b.enable_timing(True)
    for f in range(20):
        k = f % cap
        b.process_device(d_out.data_ptr() + k * esz, d_in.data_ptr() + k * esz, d_vad.data_ptr() + k * N * 4, 0, 1, st)
    torch.cuda.synchronize()
    k2 = b.kernel_ms()["network"]
    b.enable_timing(False)
    if SERIAL:
        print(f"N={N} path={path}: K2 stand-alone {k2:.4f} ms")
        b.close()
        continue
    b.process_device(d_out.data_ptr(), d_in.data_ptr(), d_vad.data_ptr(), 0, cap, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.process_device(d_out.data_ptr(), d_in.data_ptr(), d_vad.data_ptr(), 0, cap, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (10 * cap)
    print(f"N={N} path={path}: K2 stand-alone {k2:.4f} ms | pipelined {dt*1e3:.4f} ms/step = {N/dt/1e6:.2f} M frames/s")
    b.close()
