"""the torch-facing op (tensors on torch's stream) and a float-model sanity check."""
import numpy as np
import pytest

from conftest import assert_bits_equal, load_blob
from oracle.binding import Oracle
from rnnoise_amd import synth
from rnnoise_amd.torch_op import FloatNet, RNNoiseOp


def test_float_net_tracks_the_quantised_oracle():
    """independent float32/64 re-statement built from the blob: same features in, gains within the
    activation-quantisation error (1/127 steps, 3e-4 rational approximations) of the int8 network"""
    blob = load_blob("default")
    pcm = synth.stream_pcm(6, 60).astype(np.float32).reshape(60, 480)
    o = Oracle(blob)
    res = o.run(pcm)
    net = FloatNet(blob)
    err = []
    for t in range(60):
        if res["silence"][t]:
            continue
        g, v = net.step(res["features"][t].astype(np.float64))
        err.append(np.abs(g - res["gains"][t]).max())
        assert abs(v - res["vad"][t]) < 0.08
    assert max(err) < 0.08 and np.mean(err) < 0.02, (max(err), np.mean(err))


@pytest.mark.gpu
def test_op_on_torch_tensors_and_streams():
    torch = pytest.importorskip("torch")
    blob = load_blob("default")
    N, T = 6, 8
    pcm = synth.batch_pcm(range(N), T)
    op = RNNoiseOp(blob, N)
    x = torch.from_numpy(pcm).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out, vad, gains = op(x[:5])
        out2, vad2, gains2 = op(x[5:])  # state carries across calls
    s.synchronize()
    out = torch.cat([out, out2]).cpu().numpy()
    gains = torch.cat([gains, gains2]).cpu().numpy()
    for i in range(N):
        want = Oracle(blob).run(pcm[:, i])
        assert_bits_equal(out[:, i], want["out"], "pcm")
        assert_bits_equal(gains[:, i], want["gains"], "gains")


# ---- cross-check against the REFERENCE's un-quantised PyTorch model (torch/rnnoise/rnnoise.py:86-109) -------------------
# tests/golden/torch_float_default.npz = that model's forward on the checkpoint the blob was exported from, on the
# oracle's features (tests/golden/make_torch_float_golden.py).  Sanity tolerance, not bit parity: int8 weights and u8
# activations against float32; measured max |dgain| 0.016, mean 0.0033 once the start-up transient (different handling
# of the first 4 frames: causal state vs 'valid' convolutions) has died out.
TOL_MAX, TOL_MEAN, TOL_VAD, SETTLE = 0.03, 0.006, 0.015, 20


def _torch_golden():
    import os
    from conftest import GOLD
    return np.load(os.path.join(GOLD, "torch_float_default.npz"))


def test_oracle_and_float_net_track_the_reference_torch_model():
    g = _torch_golden()
    blob = load_blob("default")
    T = g["features"].shape[0]
    pcm = synth.stream_pcm(int(g["stream"]), T).astype(np.float32).reshape(T, 480)
    res = Oracle(blob).run(pcm)
    assert_bits_equal(res["features"], g["features"], "the fixture's input features")
    d = np.abs(res["gains"][SETTLE:] - g["gains"][SETTLE - 4:])
    assert d.max() < TOL_MAX and d.mean() < TOL_MEAN, (d.max(), d.mean())
    assert np.abs(res["vad"][SETTLE:] - g["vad"][SETTLE - 4:]).max() < TOL_VAD
    net = FloatNet(blob)
    fg = np.array([net.step(f.astype(np.float64))[0] for f in res["features"]])
    d = np.abs(fg[SETTLE:] - g["gains"][SETTLE - 4:])
    assert d.max() < TOL_MAX and d.mean() < TOL_MEAN, (d.max(), d.mean())


@pytest.mark.gpu
def test_registered_op_and_reference_torch_cross_check():
    """torch.ops.rnnoise_amd.process (torch.library custom op) on CUDA tensors; its raw gains against the reference's
    float PyTorch model within the sanity tolerance, and against the oracle bit for bit"""
    torch = pytest.importorskip("torch")
    g = _torch_golden()
    blob = load_blob("default")
    T = g["features"].shape[0]
    ids = [int(g["stream"])] * 16 + [1, 2]          # 18 streams: one full MFMA tile and a ragged one
    pcm = synth.batch_pcm(ids, T)
    op = RNNoiseOp(blob, len(ids))
    x = torch.from_numpy(pcm).cuda()
    out, vad, gains = torch.ops.rnnoise_amd.process(x, op.state, op.handle)
    torch.cuda.synchronize()
    assert int(op.state.item()) == T      # the tensor the op mutates (what keeps tracing from treating it as a pure function)
    gains, vad = gains.cpu().numpy(), vad.cpu().numpy()
    d = np.abs(gains[SETTLE:, 0] - g["gains"][SETTLE - 4:])
    assert d.max() < TOL_MAX and d.mean() < TOL_MEAN, (d.max(), d.mean())
    assert np.abs(vad[SETTLE:, 0] - g["vad"][SETTLE - 4:]).max() < TOL_VAD
    want = Oracle(blob).run(pcm[:, 0])
    assert_bits_equal(gains[:, 0], want["gains"], "gains")
    assert_bits_equal(gains[:, 15], want["gains"], "gains of a replica")
    assert_bits_equal(out[:, 17].cpu().numpy(), Oracle(blob).run(pcm[:, 17])["out"], "pcm of the ragged tile")
    # the fake (meta) kernel gives shapes without running anything
    with torch._subclasses.fake_tensor.FakeTensorMode():
        fo, fv, fg = torch.ops.rnnoise_amd.process(torch.empty((3, len(ids), 480), device="cuda"),
                                                   torch.zeros(1, dtype=torch.int64, device="cuda"), op.handle)
        assert fo.shape == (3, len(ids), 480) and fv.shape == (3, len(ids)) and fg.shape == (3, len(ids), 32)
    op.close()


@pytest.mark.gpu
def test_packed_model_gives_the_same_bits():
    """a batch created from the "RNPK" pack of a blob (rnnoise_amd.blob.pack) is indistinguishable from one created
    from the blob: both network paths, PCM / VAD / gains / state"""
    from rnnoise_amd import blob as rb
    from rnnoise_amd import capi
    blob = load_blob("default")
    pk = rb.pack(blob)
    pcm = synth.batch_pcm(range(20), 10, lead_silence=1)
    ref = None
    for data in (blob, pk):
        for path in (1, 0):
            m = capi.Model(data)
            b = capi.Batch(m, 20)
            b.set_nn_path(path)
            got = b.process(pcm)
            st = b.export_state(19)
            if ref is None:
                ref = (got, st)
                want = Oracle(blob).run(pcm[:, 7])
                assert_bits_equal(got[0][:, 7], want["out"], "pcm vs oracle")
            else:
                for a, c in zip(got, ref[0]):
                    assert_bits_equal(a, c, "packed vs blob")
                assert_bits_equal(st, ref[1], "state")
            b.close()
            m.close()
