"""Weight-blob tooling: read, write, inspect and synthesise "DNNw" blobs.

Format (reference: src/nnet.h:41-62 header, src/write_weights.c:46-69 writer,
src/parse_lpcnet_weights.c:37-78 reader): a stream of records, each a 64-byte header
{char head[4]="DNNw"; int32 version=0; int32 type; int32 size; int32 block_size (size rounded
up to 64); char name[44]} followed by block_size payload bytes.  Types: 0 float, 1 int,
2 qweight, 3 int8.

`synth_model()` builds a default-architecture model (conv 65x3->128 float, conv 128x3->384 int8,
3 x GRU(384) block-sparse int8 8x4 with float recurrent diagonal, dense 1536->32 / ->1 float)
from seeded random float weights with the exporter's quantisation recipe
(torch/weight-exchange/wexchange/c_export/common.py: compute_scaling :175-188, quantize_weight
:126-132, print_sparse_weight :135-171, subias/scale :245-248, dense 8x4 re-layout :59-61), so
test models of any density can be produced where the reference tree is not mounted.

CLI:  python -m rnnoise_amd.blob info  model.blob
      python -m rnnoise_amd.blob synth out.blob [--seed S] [--density D] [--boost B]
      python -m rnnoise_amd.blob pack  model.blob out.rnpk     ("DNNw" -> GPU-native "RNPK" pack, include/rnnoise_amd.h)
"""
from __future__ import annotations

import argparse
import struct
import sys
from collections import OrderedDict

import numpy as np

TYPE_FLOAT, TYPE_INT, TYPE_QWEIGHT, TYPE_INT8 = 0, 1, 2, 3
_DTYPES = {TYPE_FLOAT: np.float32, TYPE_INT: np.int32, TYPE_QWEIGHT: np.int8, TYPE_INT8: np.int8}

CONV1 = (195, 128)
CONV2 = (384, 384)
GRU = (384, 1152)
DENSE = (1536, 32)
VAD = (1536, 1)


def read_blob(blob: bytes) -> "OrderedDict[str, np.ndarray]":
    """name -> array (dtype by record type); raises ValueError on a malformed stream."""
    out, off = OrderedDict(), 0
    while off < len(blob):
        if len(blob) - off < 64:
            raise ValueError("truncated header")
        head, ver, typ, size, bs = struct.unpack_from("<4siiii", blob, off)
        name = blob[off + 20:off + 64]
        if bs < size or size <= 0 or bs > len(blob) - off - 64 or name[43] != 0:
            raise ValueError(f"bad record at byte {off}")
        name = name.split(b"\0")[0].decode()
        data = blob[off + 64:off + 64 + size]
        out[name] = np.frombuffer(data, dtype=_DTYPES.get(typ, np.uint8)).copy()
        off += 64 + bs
    return out


def write_blob(records: "OrderedDict[str, np.ndarray]") -> bytes:
    """Inverse of read_blob (the reference's write_weights: src/write_weights.c:46-69)."""
    parts = []
    for name, arr in records.items():
        arr = np.ascontiguousarray(arr)
        typ = {np.dtype(np.float32): TYPE_FLOAT, np.dtype(np.int32): TYPE_INT, np.dtype(np.int8): TYPE_INT8}[arr.dtype]
        raw = arr.tobytes()
        bs = (len(raw) + 63) // 64 * 64
        nm = name.encode()
        if len(nm) > 43:
            raise ValueError(f"name too long: {name}")
        parts.append(struct.pack("<4siiii", b"DNNw", 0, typ, len(raw), bs) + nm.ljust(44, b"\0"))
        parts.append(raw + b"\0" * (bs - len(raw)))
    return b"".join(parts)


# ---- quantisation exactly as the exporter does it -------------------------------------------
def _compute_scaling(w):  # w: (n_in, n_out)
    mx = np.max(np.abs(w), axis=0)
    ms = np.max(np.abs(w[0::2] + w[1::2]), axis=0)  # pair-sum bound: no int16 saturation in maddubs
    return np.maximum(mx / 127, ms / 129)


def _quantize(w, scale):
    q = np.round(w / (scale + 1e-30)).astype(np.int64)
    if q.max() > 127 or q.min() <= -128:
        raise ValueError("value out of bounds in quantisation")
    return q


def _linear_int8(rec, name, w, bias, sparse=False, diagonal=False):
    """w: (n_in, n_out) float.  Emits <name>_weights_int8[/_idx/_diag], _subias, _scale, _bias."""
    n_in, n_out = w.shape
    w = w.copy()
    if diagonal:  # GRU recurrent: the three gate diagonals stay in float (common.py extract_diagonal)
        diag = np.concatenate([np.diag(w[:, g * n_in:(g + 1) * n_in]).copy() for g in range(3)])
        for g in range(3):
            w[np.arange(n_in), g * n_in + np.arange(n_in)] = 0
        rec[name + "_weights_diag"] = diag.astype(np.float32)
    scale = _compute_scaling(w)
    q = _quantize(w, scale)
    if sparse:
        W, idx = [], []
        for i in range(n_out // 8):
            pos = len(idx)
            idx.append(0)
            for j in range(n_in // 4):
                blk = w[j * 4:(j + 1) * 4, i * 8:(i + 1) * 8]
                if np.sum(np.abs(blk)) > 1e-10:
                    idx.append(j * 4)
                    idx[pos] += 1
                    W.append(q[j * 4:(j + 1) * 4, i * 8:(i + 1) * 8].T.reshape(-1))  # [out8][in4]
        rec[name + "_weights_int8"] = np.concatenate(W).astype(np.int8)
        rec[name + "_weights_idx"] = np.asarray(idx, np.int32)
    else:
        v = q.reshape(n_in // 4, 4, n_out // 8, 8).transpose(2, 0, 3, 1)  # [out/8][in/4][8][4]
        rec[name + "_weights_int8"] = v.reshape(-1).astype(np.int8)
    rec[name + "_subias"] = (bias - np.sum(q * scale, axis=0)).astype(np.float32)
    rec[name + "_scale"] = (scale / 127 * np.ones(n_out)).astype(np.float32)
    rec[name + "_bias"] = bias.astype(np.float32)


def _block_sparsify(w, density, rng):
    """keep the `density` fraction of 4(in) x 8(out) blocks with the largest magnitude"""
    n_in, n_out = w.shape
    e = np.abs(w).reshape(n_in // 4, 4, n_out // 8, 8).sum(axis=(1, 3))
    k = max(1, int(round(density * e.size)))
    thr = np.sort(e.reshape(-1))[-k]
    mask = (e >= thr).astype(w.dtype)
    return w * np.repeat(np.repeat(mask, 4, axis=0), 8, axis=1)


def synth_model(seed: int = 1, density: float = 1 / 3, boost: float = 3.0) -> bytes:
    rng = np.random.Generator(np.random.PCG64(seed))
    u = lambda shape, a: rng.uniform(-a, a, size=shape)  # noqa: E731
    rec: "OrderedDict[str, np.ndarray]" = OrderedDict()
    a = boost / np.sqrt(195)
    rec["conv1_weights_float"] = u(CONV1, a).astype(np.float32).reshape(-1)
    rec["conv1_bias"] = u(128, a).astype(np.float32)
    _linear_int8(rec, "conv2", u(CONV2, boost / np.sqrt(384)), u(384, 1 / np.sqrt(384)))
    for k in (1, 2, 3):
        g = 1 / np.sqrt(384)
        wi = _block_sparsify(u(GRU, boost * g), density, rng)
        wr = u(GRU, 2 * g)
        dg = [np.diag(wr[:, j * 384:(j + 1) * 384]).copy() for j in range(3)]
        wr = _block_sparsify(wr, density, rng)
        for j in range(3):
            wr[np.arange(384), j * 384 + np.arange(384)] = dg[j]
        _linear_int8(rec, f"gru{k}_input", wi, u(1152, boost * g), sparse=True)
        _linear_int8(rec, f"gru{k}_recurrent", wr, u(1152, g), sparse=True, diagonal=True)
    rec["dense_out_weights_float"] = u(DENSE, 2 * boost / np.sqrt(1536)).astype(np.float32).reshape(-1)
    rec["dense_out_bias"] = u(32, 1.0).astype(np.float32)
    rec["vad_dense_weights_float"] = u(VAD, 2 * boost / np.sqrt(1536)).astype(np.float32).reshape(-1)
    rec["vad_dense_bias"] = u(1, 0.1).astype(np.float32)
    order = []  # same record order as the reference's rnnoise_arrays[] (weights, idx, subias, scale, bias)
    for layer, kind in (("conv1", "f"), ("conv2", "q"), ("gru1_input", "s"), ("gru1_recurrent", "d"), ("gru2_input", "s"),
                        ("gru2_recurrent", "d"), ("gru3_input", "s"), ("gru3_recurrent", "d"), ("dense_out", "f"),
                        ("vad_dense", "f")):
        if kind == "f":
            order += [layer + "_weights_float", layer + "_bias"]
        else:
            if kind == "d":
                order.append(layer + "_weights_diag")
            order.append(layer + "_weights_int8")
            if kind in "sd":
                order.append(layer + "_weights_idx")
            order += [layer + "_subias", layer + "_scale", layer + "_bias"]
    return write_blob(OrderedDict((n, rec[n]) for n in order))


def describe(blob: bytes) -> str:
    rec = read_blob(blob)
    lines = [f"{len(rec)} records, {len(blob)} bytes"]
    for layer in ("gru1_input", "gru1_recurrent", "gru2_input", "gru2_recurrent", "gru3_input", "gru3_recurrent"):
        if layer + "_weights_int8" in rec:
            nb = rec[layer + "_weights_int8"].size // 32
            lines.append(f"  {layer:<16} {nb:5d} blocks of 8x4, density {nb / (144 * 96):.3f}")
    for n, a in rec.items():
        lines.append(f"  {n:<32} {str(a.dtype):<8} {a.size:8d}")
    return "\n".join(lines)


PACK_LAYERS = ["conv1", "conv2", "gru1_input", "gru1_recurrent", "gru2_input", "gru2_recurrent", "gru3_input",
               "gru3_recurrent", "dense_out", "vad_dense"]
_PACK_HEAD = struct.Struct("<4sI8IqQ")       # magic, version, dims[8], weight_bytes, payload_bytes
_PACK_LAYER = struct.Struct("<9Q4I4i")       # 9 offsets, 4 flags, nin, nout, nblocks, pad


def pack(blob: bytes) -> bytes:
    """"DNNw" blob -> GPU-native "RNPK" pack.  The layouts are produced by the library itself (the same code that stages a
    blob for the GPU: rnnoise_amd/csrc/model.cpp stage_linear), so a pack is bit-for-bit what a blob load would upload."""
    from . import capi
    m = capi.Model(blob)
    try:
        return m.pack()
    finally:
        m.close()


def read_pack_header(data: bytes):
    """header of an "RNPK" pack: dict(version, dims, weight_bytes, payload_bytes, layers=[dict(name, nin, nout, nblocks, ...)])"""
    if data[:4] != b"RNPK" or len(data) < _PACK_HEAD.size + 10 * _PACK_LAYER.size:
        raise ValueError("not an RNPK pack")
    f = _PACK_HEAD.unpack_from(data, 0)
    layers = []
    for i, name in enumerate(PACK_LAYERS):
        v = _PACK_LAYER.unpack_from(data, _PACK_HEAD.size + i * _PACK_LAYER.size)
        layers.append(dict(name=name, offsets=dict(zip(("bias", "fw", "scale", "diag", "w", "wmf", "rowsum", "grp", "cols"), v[:9])),
                           is_int8=bool(v[12]), has_diag=bool(v[10]), nin=v[13], nout=v[14], nblocks=v[15]))
    return dict(version=f[1], dims=list(f[2:10]), weight_bytes=f[10], payload_bytes=f[11], layers=layers,
                header_bytes=_PACK_HEAD.size + 10 * _PACK_LAYER.size)


def describe_pack(data: bytes) -> str:
    h = read_pack_header(data)
    lines = [f"RNPK version {h['version']}, dims {h['dims']}, W = {h['weight_bytes']} bytes/frame, payload {h['payload_bytes']} bytes"]
    for l in h["layers"]:
        kind = f"int8, {l['nblocks']} blocks of 8x4 (+ {l['nin']}x{l['nout']} MFMA image)" if l["is_int8"] else "float"
        lines.append(f"  {l['name']:<16} {l['nin']:5d} -> {l['nout']:5d}  {kind}{', diagonal' if l['has_diag'] else ''}")
    return "\n".join(lines)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("info")
    p.add_argument("blob")
    p = sub.add_parser("synth")
    p.add_argument("out")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--density", type=float, default=1 / 3)
    p.add_argument("--boost", type=float, default=3.0)
    p = sub.add_parser("pack")
    p.add_argument("blob")
    p.add_argument("out")
    a = ap.parse_args(argv)
    if a.cmd == "info":
        data = open(a.blob, "rb").read()
        print(describe_pack(data) if data[:4] == b"RNPK" else describe(data))
    elif a.cmd == "pack":
        b = pack(open(a.blob, "rb").read())
        open(a.out, "wb").write(b)
        print(f"wrote {a.out}: {len(b)} bytes ({describe_pack(b).splitlines()[0]})")
    else:
        b = synth_model(a.seed, a.density, a.boost)
        open(a.out, "wb").write(b)
        print(f"wrote {a.out}: {len(b)} bytes")


if __name__ == "__main__":
    main(sys.argv[1:])
