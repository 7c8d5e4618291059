#!/usr/bin/env python3
"""Synthetic default-architecture model, produced with the REFERENCE's own exporter.

TEST INFRASTRUCTURE (oracle kit).  Runs only in the build container, where
/root/reference is mounted; nothing in the product imports this file.

The reference's default model (src/rnnoise_data.c/.h) is not in the tree: it is
fetched by download_model.sh:4-31 and there is no network here.  We therefore
instantiate torch/rnnoise/rnnoise.py:58-81 (RNNoise(cond_size=128, gru_size=384),
the training defaults train_rnnoise.py:48-49) with seeded random weights, force the
GRU sparsifier to its final density (rnnoise.py:43-50, gru_sparsifier.py:93-143) and
export with torch/rnnoise/dump_rnnoise_weights.py --quantize, i.e. the exact code
path that produced the real rnnoise_data.c.  The result defines (a) the compiled-in
model of oracle/_ref/librnnoise_ref_<name>.so and (b) through write_weights.c the
"DNNw" blob that the product loads.

Variants
  default : densities W_*r .3, W_*z .2, W_*n .5  (= 1/3 overall), seed 1234
  little  : half those densities (stand-in for rnnoise_data_little.c, README:121-125)
  dense   : density 1.0 everywhere (stress: largest index lists)

`--boost` multiplies a few weight tensors after init so tanh/sigmoid clamp and
saturate now and then (a plain random model keeps every gain in
[0.42,0.56] and never reaches the activation clamps).
"""
import argparse
import os
import subprocess
import sys

REF = os.environ.get("RNNOISE_REFERENCE", "/root/reference")

VARIANTS = {
    #            r     z     n    seed
    "default": (0.3, 0.2, 0.5, 1234),
    "little": (0.15, 0.1, 0.25, 4321),
    "dense": (1.0, 1.0, 1.0, 777),
}


def build(name: str, outdir: str, boost: float) -> None:
    import torch

    sys.path.insert(0, os.path.join(REF, "torch", "rnnoise"))
    import rnnoise  # the reference's model definition

    dr, dz, dn, seed = VARIANTS[name]
    for gate, d in (("r", dr), ("z", dz), ("n", dn)):
        for side in ("h", "i"):
            key = f"W_{side}{gate}"
            old = rnnoise.sparse_params1[key]
            rnnoise.sparse_params1[key] = (d, old[1], old[2])

    torch.manual_seed(seed)
    kw = {"cond_size": 128, "gru_size": 384}
    model = rnnoise.RNNoise(**kw)
    with torch.no_grad():
        if boost != 1.0:
            # widen the dynamic range of the activations (see module docstring)
            model.conv1.weight.mul_(boost)
            model.conv2.weight.mul_(boost)
            for gru in (model.gru1, model.gru2, model.gru3):
                gru.weight_ih_l0.mul_(boost)
                gru.bias_ih_l0.mul_(boost)
            model.dense_out.weight.mul_(2.0 * boost)
            model.dense_out.bias.uniform_(-1.0, 1.0)
            model.vad_dense.weight.mul_(2.0 * boost)
    for s in model.sparsifier:
        s.step_counter = s.stop
    model.sparsify()

    os.makedirs(outdir, exist_ok=True)
    pth = os.path.join(outdir, "synth.pth")
    torch.save({"model_args": (), "model_kwargs": kw, "state_dict": model.state_dict()}, pth)
    subprocess.check_call(
        [sys.executable, os.path.join(REF, "torch", "rnnoise", "dump_rnnoise_weights.py"),
         "--quantize", pth, outdir],
        stdout=subprocess.DEVNULL,
    )
    for f in ("rnnoise_data.c", "rnnoise_data.h"):
        assert os.path.exists(os.path.join(outdir, f)), f


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("name", choices=sorted(VARIANTS))
    ap.add_argument("outdir")
    ap.add_argument("--boost", type=float, default=3.0)
    a = ap.parse_args()
    build(a.name, a.outdir, a.boost)
    print(f"[gen_model] {a.name} -> {a.outdir}")
