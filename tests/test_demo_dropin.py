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
@pytest.mark.rcp("host")  # what a deployed process gets by default: the rcpps of the CPU it runs on, in the demo and in the oracle
def test_demo_output_is_byte_identical(tmp_path):
    if not os.path.exists(DEMO_BIN):
        pytest.skip("tests/_build/rnnoise_demo_dropin not built (needs the reference sources at build time)")
    from oracle.binding import Oracle
    blob = load_blob("default")
    (tmp_path / "weights_blob.bin").write_bytes(blob)
    T = 1000  # BASELINE configs[0]: 10 s of 48 kHz audio through the reference's own demo program
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


REF_DEMO = os.path.join(ROOT, "oracle", "_ref", "rnnoise_demo_ref")
REF_LIBDIR = os.path.join(ROOT, "oracle", "_ref", "soname")


def test_our_library_carries_the_reference_soname():
    """configure.ac:41-43 (libtool 4:1:4) installs librnnoise.so.0; ours must be loadable under that name"""
    so = os.path.join(ROOT, "rnnoise_amd", "librnnoise.so.0")
    assert os.path.exists(so), "build() did not produce rnnoise_amd/librnnoise.so.0"
    dyn = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "Library soname: [librnnoise.so.0]" in dyn
    nm = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    for sym in ("rnnoise_create", "rnnoise_process_frame", "rnnoise_destroy", "rnnoise_init", "rnnoise_get_size"):
        assert f" T {sym}" in nm


@pytest.mark.gpu
@pytest.mark.rcp("host")
def test_already_linked_reference_demo_runs_on_our_library(tmp_path):
    """The reference's demo, built and linked against the REFERENCE's librnnoise.so.0 (no rpath, NULL model = compiled-in
    weights), run twice: on the reference library, and with LD_LIBRARY_PATH pointing at ours plus the default-model blob.
    Same bytes out.  10 s of audio = BASELINE configs[0]."""
    if not (os.path.exists(REF_DEMO) and os.path.exists(os.path.join(REF_LIBDIR, "librnnoise.so.0"))):
        pytest.skip("oracle/_ref/rnnoise_demo_ref not built (needs the reference sources at build time)")
    blob = load_blob("default")
    (tmp_path / "default.blob").write_bytes(blob)
    T = 1000
    synth.stream_pcm(5, T, lead_silence=10).tofile(tmp_path / "in.raw")
    needed = subprocess.run(["readelf", "-d", REF_DEMO], capture_output=True, text=True).stdout
    assert "librnnoise.so.0" in needed and "RPATH" not in needed and "RUNPATH" not in needed
    base = {k: v for k, v in os.environ.items() if k != "LD_LIBRARY_PATH"}
    subprocess.check_call([REF_DEMO, "in.raw", "ref.raw"], cwd=tmp_path, timeout=300,
                          env=dict(base, LD_LIBRARY_PATH=REF_LIBDIR))
    import time
    t0 = time.perf_counter()
    subprocess.check_call([REF_DEMO, "in.raw", "ours.raw"], cwd=tmp_path, timeout=600,
                          env=dict(base, LD_LIBRARY_PATH=os.path.join(ROOT, "rnnoise_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
                                   RNNOISE_AMD_DEFAULT_MODEL=str(tmp_path / "default.blob")))
    dt = time.perf_counter() - t0
    ref = (tmp_path / "ref.raw").read_bytes()
    ours = (tmp_path / "ours.raw").read_bytes()
    # The reference's tanh / sigmoid execute THIS CPU's rcpps (src/vec_avx.h:413,442): its output is a function of the host.
    # Our library's default profile captures the same instruction at load time (rcp_profiles.h), the oracle is told to do the
    # same ("host"): all three must agree on whatever box this runs on -- Intel build host or AMD EPYC GPU box.
    from oracle import binding
    from oracle.binding import Oracle
    pcm = np.fromfile(tmp_path / "in.raw", dtype=np.int16)
    want = Oracle(blob).run(pcm.astype(np.float32).reshape(T, 480))["out"][1:].astype(np.int16).tobytes()
    assert len(ref) == (T - 1) * 480 * 2 and len(ours) == len(ref)
    assert ours == ref, "already-linked reference demo: our library's bytes differ from the reference library's on this host"
    assert ours == want, "already-linked reference demo on our library: bytes differ from the oracle"
    print(f"configs[0]: {T} frames through the pooled drop-in path in {dt:.2f} s wall ({T / dt:.0f} frames/s incl. process "
          f"start); ours == reference library == oracle on this host (rcp profile of the oracle: {binding.rcp_profile()})")
