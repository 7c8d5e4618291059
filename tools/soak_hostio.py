#!/usr/bin/env python3
"""A soak of the host-fed path on its default copy mode (two named copy engines ordered against the streams by count words and
release kernels, host_io.cpp): many calls on one batch from pinned caller memory, every ring slot reused hundreds of times.  A
missing or mis-ordered dependency between a copy and a kernel would show as a replica that differs from its neighbours (the batch is
8 distinct streams, tiled) or as a first block that leaves the oracle; both are checked after EVERY call, on the host, on what the
copy engines delivered.  int16 PCM at 65,536 streams, then float PCM at 32,768.

usage (GPU box): tools/soak_hostio.py [calls = 40] > gpurun_out/.../soak_hostio.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
import torch  # noqa: E402
from rnnoise_amd import capi, synth  # noqa: E402
from oracle.binding import Oracle  # noqa: E402
from test_gpu_parity import x86_float_to_short  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
conftest.use_rcp_profile("host")
blob = conftest.load_blob("default")
model = capi.Model(blob)
R = 8  # distinct streams
for label, N, T, s16 in (("int16 PCM, 65,536 streams", 65536, 16, True), ("float PCM, 32,768 streams", 32768, 12, False)):
    t0 = time.time()
    total = calls * T
    base = synth.batch_pcm(range(R), total, lead_silence=1)  # (total, R, 480) f32
    want = [Oracle(blob).run(base[:, s]) for s in range(R)]
    want_out = np.stack([w["out"] for w in want], axis=1)
    want_vad = np.stack([w["vad"] for w in want], axis=1)
    if s16:
        want_out = x86_float_to_short(want_out)
    dt = torch.int16 if s16 else torch.float32
    t_in = torch.empty((T, N, 480), dtype=dt).pin_memory()
    t_out = torch.empty((T, N, 480), dtype=dt).pin_memory()
    t_vad = torch.empty((T, N)).pin_memory()
    t_g = torch.empty((T, N, 32)).pin_memory()
    a_in, a_out, a_vad, a_g = t_in.numpy(), t_out.numpy(), t_vad.numpy(), t_g.numpy()
    b = capi.Batch(model, N)
    bad = 0
    for k in range(calls):
        blk = base[k * T:(k + 1) * T]
        a_in.reshape(T, N // R, R, 480)[:] = (blk.astype(np.int16) if s16 else blk)[:, None]
        a_out.fill(0)
        a_vad.fill(-1)
        b.process_into(t_out.data_ptr(), t_in.data_ptr(), t_vad.data_ptr(), t_g.data_ptr(), T, s16=s16)
        o = a_out.reshape(T, N // R, R, 480).view(np.uint16 if s16 else np.uint32)
        v = a_vad.reshape(T, N // R, R).view(np.uint32)
        g = a_g.reshape(T, N // R, R, 32).view(np.uint32)
        ok = (o == o[:, :1]).all() and (v == v[:, :1]).all() and (g == g[:, :1]).all()
        ok = ok and np.array_equal(o[:, 0], want_out[k * T:(k + 1) * T].view(o.dtype)) and np.array_equal(v[:, 0], want_vad[k * T:(k + 1) * T].view(np.uint32))
        if not ok:
            bad += 1
            print(f"{label}: call {k} differs (replicas equal: {(o == o[:, :1]).all()})", flush=True)
    print(f"{label}: {calls} calls of {T} frames from pinned memory ({total} frames per stream, {N * total:.3g} stream-frames, every ring slot reused "
          f"{total // 6} times): {'every' if not bad else 'NOT every'} call's PCM, vad and gains replica-equal and its first block bit-identical to the oracle "
          f"({bad} bad calls; {time.time() - t0:.0f} s; copy mode {os.environ.get('RNNOISE_AMD_HOSTIO_COPY', 'default = sdma')}, rcp profile {capi.rcp_profile()})", flush=True)
    b.close()
    del t_in, t_out, t_vad, t_g
    if bad:
        sys.exit(1)
