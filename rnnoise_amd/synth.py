"""Deterministic synthetic 48 kHz mono PCM for parity tests, smoke() and bench.py.

Recipe: per stream ``s`` (seed 20250223 + s)
``x = 6000*h(t) + 1500*n(t)`` rounded to int16 and handed over as float, exactly the
scaling ``examples/rnnoise_demo.c:56`` feeds ``rnnoise_process_frame`` (samples stay in
the +-32768 range).  ``h`` is a 19-harmonic voiced-like tone whose f0 wanders around
``90 + (s mod 160)`` Hz with a slow amplitude envelope, ``n`` is box-filtered Gaussian
noise.  ``lead_silence`` frames of exact zeros exercise the silence branch
(``src/denoise.c:389-393``).
"""
from __future__ import annotations

import zlib

import numpy as np

FRAME = 480
FS = 48000.0


def stream_pcm(stream: int, n_frames: int, lead_silence: int = 0) -> np.ndarray:
    """int16 PCM, shape (n_frames*480,)."""
    n = (n_frames - lead_silence) * FRAME
    out = np.zeros(n_frames * FRAME, dtype=np.int16)
    if n <= 0:
        return out
    rng = np.random.Generator(np.random.PCG64(20250223 + stream))
    t = np.arange(n, dtype=np.float64) / FS
    f0 = (90.0 + (stream % 160)) + 40.0 * np.sin(2 * np.pi * 0.5 * t)
    phi = 2 * np.pi * np.cumsum(f0) / FS
    h = np.zeros(n)
    for k in range(1, 20):
        h += np.sin(k * phi) / k
    h *= (0.5 + 0.5 * np.sin(2 * np.pi * 1.3 * t)) ** 2
    h /= max(np.max(np.abs(h)), 1e-9)
    w = rng.standard_normal(n + 7)
    nz = np.convolve(w, np.ones(8) / 8.0, mode="valid")
    x = 6000.0 * h + 1500.0 * nz
    out[lead_silence * FRAME:] = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    return out


def batch_pcm(streams, n_frames: int, lead_silence: int = 0) -> np.ndarray:
    """float32 frames, shape (n_frames, n_streams, 480) -- the layout of the batched API."""
    cols = [stream_pcm(s, n_frames, lead_silence).astype(np.float32).reshape(n_frames, FRAME) for s in streams]
    return np.ascontiguousarray(np.stack(cols, axis=1))


def crc32(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
