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
