#!/usr/bin/env python3
"""Reference-side fixture for the float cross-check of SURVEY 8f row f4 (run in the build container only).

Runs the REFERENCE's own un-quantised PyTorch model -- torch/rnnoise/rnnoise.py:86-109 (RNNoise.forward), on the
checkpoint oracle/gen_model.py exported the blob from (oracle/_ref/gen_default/synth.pth) -- on a sequence of feature
vectors produced by the oracle, and stores features in / float gains and VAD out.  The reference code is imported and
run where it lies; only its outputs are written.

  tests/golden/torch_float_default.npz:  features (T, 65), silence (T,), gains (T-4, 32), vad (T-4,)
  (two 'valid' kernel-3 convolutions: torch output k belongs to frame k+4 of the causal C implementation)

Usage:  make -C oracle ref && python tests/golden/make_torch_float_golden.py
"""
import lzma
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("RNNOISE_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "torch", "rnnoise"))
sys.path.insert(0, os.path.join(REF, "torch", "sparsification"))

import torch  # noqa: E402

import rnnoise  # noqa: E402  (the reference's model definition)
from oracle.binding import Oracle  # noqa: E402
from rnnoise_amd import synth  # noqa: E402

T = 120
blob = lzma.decompress(open(os.path.join(ROOT, "tests", "golden", "default.blob.xz"), "rb").read())
pcm = synth.stream_pcm(6, T).astype(np.float32).reshape(T, 480)
res = Oracle(blob).run(pcm)
ck = torch.load(os.path.join(ROOT, "oracle", "_ref", "gen_default", "synth.pth"), map_location="cpu")
model = rnnoise.RNNoise(*ck["model_args"], **ck["model_kwargs"])
model.load_state_dict(ck["state_dict"])
model.eval()
with torch.no_grad():
    gain, vad, _ = model(torch.from_numpy(res["features"])[None])  # (1, T, 65) -> (1, T-4, 32), (1, T-4, 1)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "torch_float_default.npz"), stream=6, features=res["features"],
                    silence=res["silence"], gains=gain[0].numpy(), vad=vad[0, :, 0].numpy())
print("wrote tests/golden/torch_float_default.npz", gain.shape, vad.shape)
