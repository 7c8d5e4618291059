"""BASELINE configs[0] (plumbing): the REFERENCE's own examples/rnnoise_demo.c, compiled
unmodified against OUR include/rnnoise.h and linked to librnnoise_amd.so, must produce the
same bytes as the demo semantics applied to the oracle (first frame dropped, truncating
(short) cast, partial tail frame dropped; examples/rnnoise_demo.c:52-61).

The binary is built where /root/reference is mounted (__graft_entry__.build()) into
tests/_build/ and travels to the GPU box with the snapshot.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_blob
from rnnoise_amd import capi, synth

DEMO_SRC = os.path.join(os.environ.get("RNNOISE_REFERENCE", "/root/reference"), "examples", "rnnoise_demo.c")
DEMO_BIN = os.path.join(ROOT, "tests", "_build", "rnnoise_demo_dropin")


def build_demo():
    os.makedirs(os.path.dirname(DEMO_BIN), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-w", "-DUSE_WEIGHTS_FILE", f"-I{ROOT}/include", DEMO_SRC, "-o", DEMO_BIN,
                           f"-L{ROOT}/rnnoise_amd", "-l:librnnoise_amd.so", f"-Wl,-rpath,{ROOT}/rnnoise_amd", "-lm"])


@pytest.mark.skipif(not os.path.exists(DEMO_SRC), reason="reference sources not mounted")
def test_reference_demo_compiles_and_links_against_our_header_and_library():
    build_demo()
    assert os.path.exists(DEMO_BIN)
    nm = subprocess.run(["nm", "-D", "--undefined-only", DEMO_BIN], capture_output=True, text=True).stdout
    for sym in ("rnnoise_create", "rnnoise_process_frame", "rnnoise_destroy", "rnnoise_model_from_filename",
                "rnnoise_model_free"):
        assert sym in nm


@pytest.mark.gpu
def test_demo_output_is_byte_identical(tmp_path):
    if not os.path.exists(DEMO_BIN):
        pytest.skip("tests/_build/rnnoise_demo_dropin not built (needs the reference sources at build time)")
    from oracle.binding import Oracle
    blob = load_blob("default")
    (tmp_path / "weights_blob.bin").write_bytes(blob)
    T = 200  # 2 s (the full 10 s case is the same code path; kept short because every frame is 3 PCIe round trips)
    pcm = synth.stream_pcm(12, T, lead_silence=4)
    raw = np.concatenate([pcm, np.arange(100, dtype=np.int16)])  # + a partial tail frame that must be dropped
    raw.tofile(tmp_path / "in.raw")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "rnnoise_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    subprocess.check_call([DEMO_BIN, "in.raw", "out.raw"], cwd=tmp_path, env=env, timeout=300)
    got = np.fromfile(tmp_path / "out.raw", dtype=np.int16)
    want = Oracle(blob).run(pcm.astype(np.float32).reshape(T, 480))["out"]
    want = want[1:].astype(np.int16).reshape(-1)  # C (short) cast truncates toward zero, like astype
    assert got.shape == want.shape
    assert np.array_equal(got, want)
