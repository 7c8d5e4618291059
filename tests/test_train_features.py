"""batched training-feature extraction = the inner loop of the reference's
src/dump_features.c:466-491 (TRAINING=1 build).  CPU: the oracle's restatement against the reference's
own TRAINING-mode functions (oracle/ref_harness_train.c).  GPU: the HIP kernel against the oracle."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from oracle.binding import RefTrainHarness, TrainOracle
from rnnoise_amd import synth


def make_case(stream, T, rng):
    clean = synth.stream_pcm(stream, T).astype(np.float32).reshape(T, 480) * 0.5
    noise = (rng.standard_normal((T, 480)) * (300 + 200 * stream)).astype(np.float32)
    noisy = clean + noise
    clean[10:14] = 0          # target silence
    noisy[10:14] = noise[10:14] * 1e-4   # ... and near-silent input: E < 0.1 branch
    vad = (np.arange(T) % 3 != 0).astype(np.float32)
    return clean, noisy, vad


@pytest.mark.skipif(not RefTrainHarness.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_train_step_matches_reference_training_build():
    rng = np.random.default_rng(5)
    for stream, (lp, blp, nf) in enumerate([(481, 32, 0), (200, 24, 0), (481, 32, 1), (90, 16, 1)]):
        clean, noisy, vad = make_case(stream, 40, rng)
        o, r = TrainOracle(), RefTrainHarness()
        for t in range(40):
            a = o.frame(clean[t], noisy[t], lp, blp, vad[t], nf)
            b = r.frame(clean[t], noisy[t], lp, blp, vad[t], nf)
            assert_bits_equal(a, b, f"stream {stream} frame {t}")
        assert (a[65:97] == -1).any() or lp == 481


@pytest.mark.gpu
def test_gpu_train_features_bit_exact():
    from conftest import load_blob
    from rnnoise_amd import capi
    rng = np.random.default_rng(6)
    cfg = [(481, 32, 0), (200, 24, 0), (481, 32, 1), (90, 16, 1), (300, 28, 0)]
    T, N = 36, len(cfg)
    cases = [make_case(s, T, rng) for s in range(N)]
    clean = np.stack([c[0] for c in cases], axis=1)
    noisy = np.stack([c[1] for c in cases], axis=1)
    vad = np.stack([c[2] for c in cases], axis=1)
    m = capi.Model(load_blob("default"))
    b = capi.Batch(m, N)
    rec = np.concatenate([b.train_features(clean[:20], noisy[:20], vad[:20], [c[0] for c in cfg], [c[1] for c in cfg],
                                           [c[2] for c in cfg]),
                          b.train_features(clean[20:], noisy[20:], vad[20:], [c[0] for c in cfg], [c[1] for c in cfg],
                                           [c[2] for c in cfg])])
    for s, (lp, blp, nf) in enumerate(cfg):
        o = TrainOracle()
        for t in range(T):
            want = o.frame(clean[t, s], noisy[t, s], lp, blp, vad[t, s], nf)
            assert_bits_equal(rec[t, s], want, f"stream {s} frame {t}")
