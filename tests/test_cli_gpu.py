"""the multi-stream file front end reproduces, per file, the byte stream the
reference demo would write (examples/rnnoise_demo.c:52-61) -- checked against the oracle."""
import numpy as np
import pytest

from conftest import load_blob
from oracle.binding import Oracle
from rnnoise_amd import cli, synth

pytestmark = pytest.mark.gpu


def test_three_files_of_different_lengths(tmp_path):
    blob = load_blob("default")
    lens = [130 * 480 + 77, 40 * 480, 201 * 480 + 479]  # with partial tails that must be ignored
    paths = []
    for s, n in enumerate(lens):
        x = synth.stream_pcm(20 + s, n // 480 + 1, lead_silence=2)[:n]
        p = tmp_path / f"in{s}.raw"
        x.tofile(p)
        paths.append(str(p))
    (tmp_path / "w.blob").write_bytes(blob)
    cli.main(["denoise", "--model", str(tmp_path / "w.blob"), "--out-dir", str(tmp_path / "out"), "--chunk-frames", "64",
              "--vad-csv"] + paths)
    for s, n in enumerate(lens):
        T = n // 480
        x = np.fromfile(paths[s], dtype=np.int16)[:T * 480].astype(np.float32).reshape(T, 480)
        want = Oracle(blob).run(x)
        got = np.fromfile(tmp_path / "out" / f"in{s}.raw.denoised.raw", dtype=np.int16)
        assert got.size == (T - 1) * 480
        assert np.array_equal(got, want["out"][1:].astype(np.int16).reshape(-1))
        vad = np.loadtxt(tmp_path / "out" / f"in{s}.raw.vad.csv")
        assert np.allclose(vad, want["vad"], atol=5e-7)
