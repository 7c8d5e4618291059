#!/usr/bin/env python3
"""Diagnostic: N streams = replicas of a 32-stream block through pipelined calls; reports where replicas diverge.
usage: tools/diag_replicas.py [N=32768] [model=little] [reps=6]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rnnoise_amd import capi, synth
from conftest import load_blob
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
blob = load_blob(sys.argv[2] if len(sys.argv) > 2 else "little")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
capi.set_rcp_profile("intel")
calls = (5, 1, 8); T = sum(calls)
base = synth.batch_pcm(range(32), T); base[:3, 9] = 0; base[T - 5:T - 3, 12] = 0
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(base).to(dev).repeat(1, N // 32, 1).contiguous()
m = capi.Model(blob)
for rep in range(reps):
    d_out = torch.empty_like(d_in); d_vad = torch.empty((T, N), device=dev); d_gains = torch.empty((T, N, 32), device=dev)
    b = capi.Batch(m, N); b.set_nn_path(1)
    st = torch.cuda.current_stream().cuda_stream; f = 0
    for n in calls:
        b.process_device(d_out[f].data_ptr(), d_in[f].data_ptr(), d_vad[f].data_ptr(), d_gains[f].data_ptr(), n, st); f += n
    torch.cuda.synchronize()
    msg = []
    for name, t, w in (("gains", d_gains, 32), ("vad", d_vad, 1), ("pcm", d_out, 480)):
        r = t.view(torch.int32).reshape(T, N // 32, 32, w)
        bad = (r != r[:, :1]).any(dim=3)          # [T, replicas, 32]
        if bool(bad.any()):
            idx = bad.nonzero()
            fr = sorted(set(idx[:, 0].tolist())); rp = sorted(set(idx[:, 1].tolist())); ss = sorted(set(idx[:, 2].tolist()))
            first = idx[0].tolist()
            msg.append(f"{name}: {idx.shape[0]} (frame,replica,stream) differ; frames {fr[:8]} replicas {rp[:12]}{'...' if len(rp) > 12 else ''} (n={len(rp)}) streams {ss[:16]} first {first}")
    print(f"rep {rep}: " + ("; ".join(msg) if msg else "all replicas identical"), flush=True)
    b.close()
