"""Blob tooling: rnnoise_amd/blob.py reads/writes the reference's "DNNw" format and
synthesises default-architecture models that the oracle, the library and (where built) the
reference itself all accept."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from oracle.binding import Oracle, RefHarness
from rnnoise_amd import blob as rb
from rnnoise_amd import capi, synth


def test_read_write_round_trip_is_byte_identical(blob_default, blob_little):
    for b in (blob_default, blob_little):
        rec = rb.read_blob(b)
        assert len(rec) == 43 and list(rec)[0] == "conv1_weights_float"
        assert rb.write_blob(rec) == b  # the reference writer's exact bytes (src/write_weights.c:46-69)


def test_malformed_streams_raise():
    with pytest.raises(ValueError):
        rb.read_blob(b"\0" * 63)
    with pytest.raises(ValueError):
        rb.read_blob(b"DNNw" + b"\0" * 60)  # size 0


@pytest.mark.parametrize("density", [0.1, 1 / 3, 1.0])
def test_synthetic_models_are_accepted_and_behave(density):
    b = rb.synth_model(seed=7, density=density)
    rec = rb.read_blob(b)
    assert len(rec) == 43
    nb = rec["gru2_input_weights_int8"].size // 32
    assert abs(nb / (144 * 96) - density) < 0.02
    m = capi.Model(b)  # library-side parser + weight-byte accounting (no GPU needed)
    blocks = sum(rec[f"gru{k}_{side}_weights_int8"].size // 32 for k in (1, 2, 3) for side in ("input", "recurrent"))
    expect_w = 303236 + 150528 + 32 * blocks + 4 * (blocks + 6 * 144) + 6 * 9216 + 3 * 4608  # SURVEY 8d formula
    assert m.weight_bytes == expect_w
    pcm = synth.stream_pcm(1, 40).astype(np.float32).reshape(40, 480)
    res = Oracle(b).run(pcm)
    assert np.isfinite(res["out"]).all() and res["gains"].std() > 0.05
    if RefHarness.available():  # the reference's own parser and kernels agree with the oracle on it
        ref = RefHarness(b).run(pcm)
        for k in ("out", "gains", "vad", "pitch"):
            assert_bits_equal(res[k], ref[k], k)


def test_cli_info(tmp_path, capsys, blob_default):
    p = tmp_path / "m.blob"
    p.write_bytes(blob_default)
    rb.main(["info", str(p)])
    out = capsys.readouterr().out
    assert "43 records" in out and "gru3_recurrent" in out and "density 0.33" in out
    rb.main(["synth", str(tmp_path / "s.blob"), "--seed", "3", "--density", "0.2"])
    assert len(rb.read_blob((tmp_path / "s.blob").read_bytes())) == 43


@pytest.mark.gpu
@pytest.mark.parametrize("density,path", [(1.0, 1), (1.0, 0), (0.1, 1), (0.1, 0)])
def test_synthetic_models_on_gpu(density, path):
    """dense (largest index lists, SURVEY config stress) and very sparse models (short, ragged block lists: chunks of the
    row-major copy padded with zero blocks), both network paths"""
    b = rb.synth_model(seed=11, density=density)
    m = capi.Model(b)
    pcm = synth.batch_pcm([0, 5, 9], 30, lead_silence=2)
    batch = capi.Batch(m, 3)
    batch.set_nn_path(path)
    out, vad, gains = batch.process(pcm)
    for i in range(3):
        want = Oracle(b).run(pcm[:, i])
        assert_bits_equal(gains[:, i], want["gains"], "gains")
        assert_bits_equal(vad[:, i], want["vad"], "vad")
        assert_bits_equal(out[:, i], want["out"], "pcm")


# ---- "RNPK": the GPU-native packed model (include/rnnoise_amd.h rnnoise_amd_model_pack) --------------
def test_pack_round_trip_and_header(blob_default, blob_little, tmp_path):
    from rnnoise_amd import capi
    for b in (blob_default, blob_little):
        p = rb.pack(b)
        h = rb.read_pack_header(p)
        assert p[:4] == b"RNPK" and h["version"] == 2 and h["dims"] == [195, 128, 384, 384, 384, 1536, 32, 64]
        assert h["weight_bytes"] == capi.Model(b).weight_bytes and h["payload_bytes"] == len(p) - h["header_bytes"]
        assert [l["name"] for l in h["layers"]] == rb.PACK_LAYERS
        m = capi.Model(p)                       # rnnoise_model_from_buffer accepts the pack
        assert m.weight_bytes == h["weight_bytes"]
        assert m.pack() == p                    # packing a packed model reproduces it byte for byte
    # the int8 payload of a layer is the blob's own block stream, untouched
    rec = rb.read_blob(blob_default)
    h = rb.read_pack_header(rb.pack(blob_default))
    l = h["layers"][2]
    w = np.frombuffer(rb.pack(blob_default), np.int8, 32 * l["nblocks"], h["header_bytes"] + l["offsets"]["w"])
    assert np.array_equal(w, rec["gru1_input_weights_int8"])
    # command-line form + loading through the file entry point
    (tmp_path / "m.blob").write_bytes(blob_default)
    rb.main(["pack", str(tmp_path / "m.blob"), str(tmp_path / "m.rnpk")])
    assert (tmp_path / "m.rnpk").read_bytes() == rb.pack(blob_default)
    L = capi.lib()
    mh = L.rnnoise_model_from_filename(str(tmp_path / "m.rnpk").encode())
    assert mh and L.rnnoise_model_weight_bytes(mh) == h["weight_bytes"]
    L.rnnoise_model_free(mh)


def test_corrupt_packs_are_rejected(blob_default):
    import struct
    from rnnoise_amd import capi
    good = bytearray(rb.pack(blob_default))
    h = rb.read_pack_header(bytes(good))

    def rejected(b):
        try:
            capi.Model(bytes(b)).weight_bytes
        except ValueError:
            return True
        return False

    assert not rejected(good)
    bad = bytearray(good); struct.pack_into("<I", bad, 4, 1); assert rejected(bad)                      # unknown version
    bad = bytearray(good); struct.pack_into("<I", bad, 8 + 4 * 4, 512); assert rejected(bad)            # other GRU size
    assert rejected(good[:-1]) and rejected(good + b"\\0")                                               # payload size mismatch
    assert rejected(good[:200])                                                                         # truncated header
    lay = 56 + 2 * 104                                                                                  # gru1_input record
    bad = bytearray(good); struct.pack_into("<Q", bad, lay + 5 * 8, h["payload_bytes"] - 16); assert rejected(bad)   # MFMA image runs off the end
    bad = bytearray(good); struct.pack_into("<i", bad, lay + 9 * 8 + 16 + 8, 10 ** 6); assert rejected(bad)          # absurd block count
    cols = h["header_bytes"] + h["layers"][2]["offsets"]["cols"]
    bad = bytearray(good); struct.pack_into("<H", bad, cols, 382); assert rejected(bad)                 # column index past the input
    # the four kind flags (u32 has_fw, has_diag, has_cols, is_int8 after the nine offsets) are the architecture's, not the
    # file's: flipping any of them on any layer is a rejection, not a null pointer or an out-of-bounds read on the device
    for layer in range(10):
        for flag in range(4):
            off = 56 + layer * 104 + 9 * 8 + 4 * flag
            bad = bytearray(good)
            struct.pack_into("<I", bad, off, 0 if struct.unpack_from("<I", good, off)[0] else 1)
            assert rejected(bad), (layer, flag)
    lay1 = 56 + 1 * 104                                                                                 # conv2: dense int8
    bad = bytearray(good); struct.pack_into("<i", bad, lay1 + 9 * 8 + 16 + 8, 7); assert rejected(bad)  # a dense layer with 7 blocks
