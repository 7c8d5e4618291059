#!/usr/bin/env python3
"""tools/soak_long.py for the batch sizes BELOW the named ones, where round 6's last day moved the dispatch switches: the soak of
tests/test_gpu_at_size.py::_soak (calls of 8 + 5 + 1 + 8 + 2 frames: pipelined calls and one-frame calls alternate, every replica compared
with replica 0 after every call, the first block and four states with the oracle at the end) at 2,560 streams (the smallest batch on the
four-stream analysis workgroups), 4,096 (configs[1]: eight-wave tile kernel in the pipelined calls, sixteen-wave in the one-frame calls),
8,192 and 10,208 (the tile kernel at two and three tiles per CU), 10,240 and 12,288 (the layer-wise network's eight-wave layer kernel).

usage (GPU box): tools/soak_mid.py [cycles of 24 frames = 417] > gpurun_out/.../soak_mid.txt"""
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
conftest.use_rcp_profile("host")
blob = conftest.load_blob("default")
for n in (2560, 4096, 8192, 10208, 10240, 12288):
    t0 = time.time()
    t._soak(blob, n, reps=1, cycles=cycles)
    print(f"default model, {n:,} streams: {cycles * 24} frames = {n * cycles * 24:.3g} stream-frames: every replica equal to replica 0 after every call, "
          f"first block and four states bit-identical to the oracle ({time.time() - t0:.0f} s; rcp profile {capi.rcp_profile()}, log10 model {capi.log10_model()})", flush=True)
