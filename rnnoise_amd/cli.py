"""Multi-stream file front end: N RAW s16 48 kHz mono files in, N denoised
files out, one GPU batch.

Per file the semantics are those of the reference's examples/rnnoise_demo.c:52-61: samples are
fed unscaled (+-32768 range), the first output frame is dropped, the float result is cast to
short by truncation, a trailing partial frame is ignored.  The samples cross PCIe as int16 both
ways (rnnoise_batch_process_s16: the two conversions of rnnoise_demo.c:56,58 run on the device).  Files of different lengths share the
batch; a stream whose file has ended is fed zeros and produces no more output.

  python -m rnnoise_amd.cli denoise --model weights_blob.bin --out-dir out  a.raw b.raw ...
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import capi

FRAME = capi.FRAME


def denoise_files(model_blob: bytes, inputs, out_dir: str, chunk_frames: int = 100, device: int = 0,
                  vad_csv: bool = False):
    """Streams the files through the batch chunk by chunk: at most `chunk_frames` frames of every file are in host memory
    at a time (two staging buffers, reused), whatever the file lengths."""
    os.makedirs(out_dir, exist_ok=True)
    n_frames = [os.path.getsize(p) // 2 // FRAME for p in inputs]  # partial tail dropped (rnnoise_demo.c:55)
    N, T = len(inputs), max(n_frames + [0])
    model = capi.Model(model_blob)
    batch = capi.Batch(model, N, device=device)
    ins = [open(p, "rb") for p in inputs]
    outs = [open(os.path.join(out_dir, os.path.basename(p) + ".denoised.raw"), "wb") for p in inputs]
    vfs = [open(os.path.join(out_dir, os.path.basename(p) + ".vad.csv"), "w") for p in inputs] if vad_csv else None
    buf = np.zeros((min(chunk_frames, max(T, 1)), N, FRAME), np.int16)
    for t0 in range(0, T, chunk_frames):
        tn = min(chunk_frames, T - t0)
        chunk = buf[:tn]
        chunk[:] = 0  # a stream whose file has ended is fed zeros and produces no more output
        for s, f in enumerate(ins):
            k = max(0, min(tn, n_frames[s] - t0))
            if k:
                x = np.frombuffer(f.read(k * FRAME * 2), dtype=np.int16)
                chunk[:k, s] = x.reshape(k, FRAME)
        out, vad, _ = batch.process_s16(chunk, want_gains=False)
        for s in range(N):
            k = max(0, min(tn, n_frames[s] - t0))
            first = 1 if t0 == 0 else 0  # the demo drops the first output frame (rnnoise_demo.c:59-60)
            if k > first:
                outs[s].write(out[first:k, s].tobytes())  # (the demo's truncating (short) cast was done on the device)
            if vfs and k:
                vfs[s].write("".join(f"{v:.6f}\n" for v in vad[:k, s]))
    for f in ins + outs + (vfs or []):
        f.close()
    batch.close()
    model.close()
    return n_frames


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("denoise")
    p.add_argument("--model", required=True, help='"DNNw" weight blob (src/write_weights.c format)')
    p.add_argument("--out-dir", required=True)
    p.add_argument("--chunk-frames", type=int, default=100)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--vad-csv", action="store_true")
    p.add_argument("inputs", nargs="+")
    a = ap.parse_args(argv)
    n = denoise_files(open(a.model, "rb").read(), a.inputs, a.out_dir, a.chunk_frames, a.device, a.vad_csv)
    print(f"denoised {len(a.inputs)} streams, {sum(n)} frames")


if __name__ == "__main__":
    main(sys.argv[1:])
