#!/usr/bin/env python3
"""Generate the golden fixtures from the REFERENCE ITSELF (run in the build container only).

The reference has no tests and no golden vectors, so the pin is: the
reference's own sources, compiled unmodified by oracle/Makefile (pinned build, SURVEY fact
8) on this host, driven through oracle/ref_harness.c with the synthetic default-architecture
model exported by the reference's own exporter (oracle/gen_model.py).  Everything written
here is reference OUTPUT; no reference code is involved in producing the files other than by
running it.

  tests/golden/default.blob.xz   -- the "DNNw" weight blob (src/write_weights.c format)
  tests/golden/little.blob.xz    -- higher-sparsity variant (rnnoise_data_little stand-in)
  tests/golden/detail_default.npz-- 2 streams x 100 frames, everything per frame + final state
  tests/golden/digest_default.npz-- 4 streams x 400 frames: vad, pitch, raw gains, CRC32 of PCM out
  tests/golden/edge_default.npz  -- loud noise / DC / impulses / gaps, 60 frames each
  tests/golden/digest_little.npz -- 2 streams x 200 frames on the sparser model

The reference's tanh / sigmoid execute the generating CPU's `rcpps` (src/vec_avx.h:413,442), so a fixture belongs to one
CPU family and says which ("rcp_profile", "host_cpu" entries).  The Intel build host writes tests/golden/*.npz; run on the
GPU boxes' AMD EPYC host (the compiled reference under oracle/_ref travels there) the same script writes the same streams
into tests/golden/amd_zen5/ -- `python tests/golden/make_golden.py [OUTDIR]`.

Usage:  make -C oracle ref && python tests/golden/make_golden.py
"""
import lzma
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding  # noqa: E402
from oracle.binding import RefHarness  # noqa: E402
from rnnoise_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def crc_rows(a):
    return np.array([zlib.crc32(np.ascontiguousarray(r).tobytes()) & 0xFFFFFFFF for r in a], np.uint32)


def edge_inputs():
    rng = np.random.Generator(np.random.PCG64(99))
    T = 60
    loud = rng.integers(-32768, 32767, size=(T, 480)).astype(np.int16)
    dc = np.full((T, 480), 12000, np.int16)
    imp = np.zeros((T, 480), np.int16)
    imp[::7, 13] = 30000
    gaps = synth.stream_pcm(5, T).reshape(T, 480).copy()
    gaps[20:35] = 0
    gaps[45:50] //= 1000
    return dict(loud=loud, dc=dc, impulses=imp, gaps=gaps)


def host_profile():
    """The reference executes this CPU's rcpps (src/vec_avx.h:413,442): its outputs belong to one CPU family, and every
    fixture says which (rnnoise_amd/csrc/rcp_profiles.h).  A host whose table is neither built-in one cannot make goldens."""
    binding.set_rcp_profile("host")
    name = binding.rcp_profile()
    assert name in ("intel", "amd-zen5"), "this CPU's rcpps matches no committed profile: capture it first (oracle/rcp_capture.c)"
    cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
    return dict(rcp_profile=np.array(name), host_cpu=np.array(cpu))


def main():
    global GOLD
    tag = host_profile()
    blob_dir = GOLD
    if str(tag["rcp_profile"]) != "intel":
        GOLD = os.path.join(GOLD, str(tag["rcp_profile"]).replace("-", "_"))
    if len(sys.argv) > 1:
        GOLD = sys.argv[1]
    os.makedirs(GOLD, exist_ok=True)
    for name in ("default", "little"):
        blob = open(os.path.join(ROOT, "oracle", "_ref", f"{name}.blob"), "rb").read()
        path = os.path.join(blob_dir, f"{name}.blob.xz")
        if os.path.exists(path) and lzma.decompress(open(path, "rb").read()) == blob:
            continue  # unchanged: keep the committed bytes
        with open(path, "wb") as f:
            f.write(lzma.compress(blob, preset=9))
    blob = open(os.path.join(ROOT, "oracle", "_ref", "default.blob"), "rb").read()

    # ---- detail ----
    d = {}
    for s in (3, 77):
        T = 100
        pcm = synth.stream_pcm(s, T, lead_silence=12).reshape(T, 480)
        r = RefHarness(blob)
        res = r.run(pcm.astype(np.float32))
        d[f"s{s}_pcm"] = pcm
        for k, v in res.items():
            d[f"s{s}_{k}"] = v
        d[f"s{s}_state"] = r.get_state()
    np.savez_compressed(os.path.join(GOLD, "detail_default.npz"), **d, **tag)

    # ---- digest ----
    d = {}
    for s in (0, 1, 159, 4095):
        T = 400
        pcm = synth.stream_pcm(s, T, lead_silence=5).reshape(T, 480)
        r = RefHarness(blob)
        res = r.run(pcm.astype(np.float32))
        d[f"s{s}_pcm_crc"] = np.uint32(synth.crc32(pcm))
        d[f"s{s}_vad"] = res["vad"]
        d[f"s{s}_pitch"] = res["pitch"].astype(np.int16)
        d[f"s{s}_gains"] = res["gains"]
        d[f"s{s}_out_crc"] = crc_rows(res["out"])
        d[f"s{s}_state_crc"] = np.uint32(synth.crc32(r.get_state()))
    np.savez_compressed(os.path.join(GOLD, "digest_default.npz"), **d, **tag)

    # ---- edge cases ----
    d = {}
    for name, pcm in edge_inputs().items():
        r = RefHarness(blob)
        res = r.run(pcm.astype(np.float32))
        d[f"{name}_pcm"] = pcm
        d[f"{name}_vad"] = res["vad"]
        d[f"{name}_pitch"] = res["pitch"].astype(np.int16)
        d[f"{name}_gains"] = res["gains"]
        d[f"{name}_silence"] = res["silence"].astype(np.int8)
        d[f"{name}_out_crc"] = crc_rows(res["out"])
        d[f"{name}_state_crc"] = np.uint32(synth.crc32(r.get_state()))
    np.savez_compressed(os.path.join(GOLD, "edge_default.npz"), **d, **tag)

    # ---- sparser model ----
    blob2 = open(os.path.join(ROOT, "oracle", "_ref", "little.blob"), "rb").read()
    d = {}
    for s in (2, 31):
        T = 200
        pcm = synth.stream_pcm(s, T, lead_silence=3).reshape(T, 480)
        r = RefHarness(blob2)
        res = r.run(pcm.astype(np.float32))
        d[f"s{s}_pcm_crc"] = np.uint32(synth.crc32(pcm))
        d[f"s{s}_vad"] = res["vad"]
        d[f"s{s}_pitch"] = res["pitch"].astype(np.int16)
        d[f"s{s}_gains"] = res["gains"]
        d[f"s{s}_out_crc"] = crc_rows(res["out"])
        d[f"s{s}_state_crc"] = np.uint32(synth.crc32(r.get_state()))
    np.savez_compressed(os.path.join(GOLD, "digest_little.npz"), **d, **tag)
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
