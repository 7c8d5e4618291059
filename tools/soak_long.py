#!/usr/bin/env python3
"""A long run of the at-size soak test (tests/test_gpu_at_size.py::_soak): the pipelined schedule at 65,536 streams (default model) and at
32,768 streams (sparser blob) for ~10,000 frames each -- 6.6e8 and 3.3e8 stream-frames -- with every replica compared with replica 0 on the
GPU after every call and the first 32-stream block compared with the oracle frame by frame at the end.

usage (GPU box): tools/soak_long.py [cycles of 24 frames = 417] [configurations to run = 2] > gpurun_out/.../soak_long.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401
import test_gpu_at_size as t  # noqa: E402
from rnnoise_amd import capi  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 417
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
conftest.use_rcp_profile("host")
for name, blob, n in (("default model, 65,536 streams", conftest.load_blob("default"), 65536), ("sparser blob, 32,768 streams", conftest.load_blob("little"), 32768))[:n_cfg]:
    t0 = time.time()
    t._soak(blob, n, reps=1, cycles=cycles)
    print(f"{name}: {cycles * 24} frames in calls of 8 + 5 + 1 + 8 + 2 on the default schedule = {n * cycles * 24:.3g} stream-frames: every replica equal to "
          f"replica 0 after every call, first block and four states bit-identical to the oracle ({time.time() - t0:.0f} s; rcp profile {capi.rcp_profile()}, "
          f"log10 model {capi.log10_model()}, layer kernel default)", flush=True)
