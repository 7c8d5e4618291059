"""GPU suite: the HIP path, called through the C-ABI, against the oracle and the goldens.

Bar (BASELINE.json north_star: gains within 1e-4 relative): we demand MORE -- bit-identical
features, pitch, raw gains, VAD, PCM and exported state -- because the int8 quantisers turn
1-ULP deviations into >1e-4 gain deviations.  The one tolerated exception is
documented in DESIGN.md: `log10` (double, libm on the CPU vs ocml on the GPU) may round a
feature differently with probability ~1e-9 per call; the tolerance fallback below (1e-4
relative on gains, stated by north_star) is reported, never silently used.
"""
import zlib

import numpy as np
import pytest

from conftest import both_profiles, assert_bits_equal, golden
from oracle.binding import Oracle
from rnnoise_amd import capi, synth

pytestmark = pytest.mark.gpu

TOL_GAIN_REL = 1e-4  # north_star tolerance, only consulted when bit-identity fails


def crc_rows(a):
    return np.array([zlib.crc32(np.ascontiguousarray(r).tobytes()) & 0xFFFFFFFF for r in a], np.uint32)


def oracle_run(blob, pcm_TNF, collect_state=True):
    """per-stream oracle over a (T, N, 480) batch"""
    T, N, _ = pcm_TNF.shape
    outs = dict(out=np.zeros_like(pcm_TNF), vad=np.zeros((T, N), np.float32), gains=np.zeros((T, N, 32), np.float32),
                features=np.zeros((T, N, 65), np.float32), pitch=np.zeros((T, N), np.int32),
                silence=np.zeros((T, N), np.int32), state=np.zeros((N, capi.STATE_FLOATS), np.float32))
    for s in range(N):
        o = Oracle(blob)
        r = o.run(pcm_TNF[:, s])
        for k in ("out", "vad", "gains", "features", "pitch", "silence"):
            outs[k][:, s] = r[k]
        outs["state"][s] = o.get_state()
    return outs


def gpu_run(batch, pcm_TNF, per_frame_debug=False):
    T, N, _ = pcm_TNF.shape
    if not per_frame_debug:
        out, vad, gains = batch.process(pcm_TNF)
        return dict(out=out, vad=vad, gains=gains)
    res = dict(out=np.zeros_like(pcm_TNF), vad=np.zeros((T, N), np.float32), gains=np.zeros((T, N, 32), np.float32),
               features=np.zeros((T, N, 65), np.float32), pitch=np.zeros((T, N), np.int32),
               silence=np.zeros((T, N), np.int32))
    for t in range(T):
        o, v, g = batch.process(pcm_TNF[t:t + 1])
        f, s, p = batch.debug_last()
        res["out"][t], res["vad"][t], res["gains"][t] = o[0], v[0], g[0]
        res["features"][t], res["silence"][t], res["pitch"][t] = f, s, p
    return res


@pytest.fixture(scope="module")
def model(blob_default):
    return capi.Model(blob_default)


def test_device_present():
    assert capi.lib().rnnoise_amd_device_count() >= 1


SECTIONS = [("xlp", 0, 864), ("ac", 864, 869), ("lpc2", 869, 874), ("xcorr_coarse", 880, 1027), ("best", 1030, 1036),
            ("xcorr_fine", 1040, 1334), ("doubling", 1340, 1347)]


def test_pitch_stage_taps(blob_default):
    """teacher-forced pitch analysis: every intermediate of rnn_pitch_downsample / rnn_pitch_search /
    rnn_remove_doubling against the oracle, fed with the GPU's own pitch buffer.  The taps exist in the instrumented build
    of the kernels only (librnnoise_amd_instr.so, include/rnnoise_amd_debug.h): same sources, -DRN_INSTRUMENT=1."""
    with capi.instrumented():
        _pitch_stage_taps(capi.Model(blob_default), blob_default)


@pytest.mark.parametrize("n", [3, 70])
def test_instrumented_build_computes_the_same_bits(blob_default, n):
    """the instrumented library is the same code with taps: same outputs as the product library (70 streams: MFMA tiles)"""
    T = 12
    pcm = synth.batch_pcm([(7 * s) % 9 for s in range(n)], T, lead_silence=1)
    a = capi.Batch(capi.Model(blob_default), n).process(pcm)
    with capi.instrumented():
        m = capi.Model(blob_default)
        b = capi.Batch(m, n)
        b.debug_pitch(arm_only=True)
        c = b.process(pcm)
        b.close()
        m.close()
    for x, y, name in zip(a, c, ("pcm", "vad", "gains")):
        assert_bits_equal(x, y, name)


def _pitch_stage_taps(model, blob_default):
    streams = [3, 8]
    T = 14
    pcm = synth.batch_pcm(streams, T, lead_silence=2)
    b = capi.Batch(model, len(streams))
    b.debug_pitch(arm_only=True)
    orc = [Oracle(blob_default) for _ in streams]
    prev = [(0, 0.0)] * len(streams)
    for t in range(T):
        b.process(pcm[t:t + 1])
        taps = b.debug_pitch()
        for i in range(len(streams)):
            orc[i].process(pcm[t, i])
            st = b.export_state(i)
            ost = orc[i].get_state()
            assert_bits_equal(st[960:960 + 1728], ost[960:960 + 1728], f"frame {t} pitch_buf (biquad)")
            Tp, gain, want = Oracle.pitch_debug(st[960:960 + 1728], prev[i][0], prev[i][1])
            for name, a0, a1 in SECTIONS:
                assert_bits_equal(taps[i, a0:a1], want[a0:a1], f"frame {t} stream {i} {name}")
            assert int(st[960 + 1728 + 1:960 + 1728 + 2].view(np.int32)[0]) == Tp
            prev[i] = (Tp, gain)


def test_free_running_bit_exact_with_stage_taps(model, blob_default):
    """4 streams x 60 frames incl. leading silence: every stage output identical to the oracle"""
    streams = [3, 8, 77, 130]
    T = 60
    pcm = synth.batch_pcm(streams, T, lead_silence=6)
    want = oracle_run(blob_default, pcm)
    b = capi.Batch(model, len(streams))
    got = gpu_run(b, pcm, per_frame_debug=True)
    assert np.array_equal(got["silence"], want["silence"])
    assert np.array_equal(got["pitch"], want["pitch"]), np.argwhere(got["pitch"] != want["pitch"])[:5]
    assert_bits_equal(got["features"], want["features"], "features")
    assert_bits_equal(got["gains"], want["gains"], "raw gains")
    assert_bits_equal(got["vad"], want["vad"], "vad")
    assert_bits_equal(got["out"], want["out"], "pcm")
    for i in range(len(streams)):
        assert_bits_equal(b.export_state(i), want["state"][i], f"state of stream {i}")
    assert want["silence"][:6].all() and not want["silence"][8:].any()


@both_profiles
def test_detail_golden_from_reference(model):
    """reference outputs recorded in the build container (tests/golden/make_golden.py)"""
    g = golden("detail_default.npz")
    pcm = np.stack([g["s3_pcm"], g["s77_pcm"]], axis=1).astype(np.float32)
    b = capi.Batch(model, 2)
    got = gpu_run(b, pcm, per_frame_debug=True)
    for i, s in enumerate((3, 77)):
        assert np.array_equal(got["pitch"][:, i], g[f"s{s}_pitch"])
        assert np.array_equal(got["silence"][:, i], g[f"s{s}_silence"])
        assert_bits_equal(got["features"][:, i], g[f"s{s}_features"], "features")
        assert_bits_equal(got["gains"][:, i], g[f"s{s}_gains"], "gains")
        assert_bits_equal(got["vad"][:, i], g[f"s{s}_vad"], "vad")
        assert_bits_equal(got["out"][:, i], g[f"s{s}_out"], "pcm")
        assert_bits_equal(b.export_state(i), g[f"s{s}_state"], "state")


@both_profiles
def test_digest_golden_400_frames(model):
    g = golden("digest_default.npz")
    streams = (0, 1, 159, 4095)
    pcm = synth.batch_pcm(streams, 400, lead_silence=5)
    b = capi.Batch(model, 4)
    got = gpu_run(b, pcm)
    for i, s in enumerate(streams):
        assert synth.crc32(pcm[:, i].astype(np.int16)) == int(g[f"s{s}_pcm_crc"])
        ok = np.array_equal(got["gains"][:, i].view(np.uint32), g[f"s{s}_gains"].view(np.uint32))
        if not ok:  # report against the stated tolerance instead of hiding it
            rel = np.abs(got["gains"][:, i] - g[f"s{s}_gains"]) / np.maximum(np.abs(g[f"s{s}_gains"]), 1e-9)
            raise AssertionError(f"stream {s}: gains not bit-identical; max rel {rel.max():.3e} "
                                 f"(north_star tolerance {TOL_GAIN_REL}), frames over: {(rel.max(1) > TOL_GAIN_REL).sum()}")
        assert_bits_equal(got["vad"][:, i], g[f"s{s}_vad"], "vad")
        assert np.array_equal(crc_rows(got["out"][:, i]), g[f"s{s}_out_crc"])
        assert synth.crc32(b.export_state(i)) == int(g[f"s{s}_state_crc"])


@both_profiles
def test_edge_case_golden(model):
    g = golden("edge_default.npz")
    names = ["loud", "dc", "impulses", "gaps"]
    pcm = np.stack([g[f"{n}_pcm"] for n in names], axis=1).astype(np.float32)
    b = capi.Batch(model, 4)
    got = gpu_run(b, pcm)
    for i, n in enumerate(names):
        assert_bits_equal(got["gains"][:, i], g[f"{n}_gains"], f"{n} gains")
        assert_bits_equal(got["vad"][:, i], g[f"{n}_vad"], f"{n} vad")
        assert np.array_equal(crc_rows(got["out"][:, i]), g[f"{n}_out_crc"]), n
        assert synth.crc32(b.export_state(i)) == int(g[f"{n}_state_crc"]), n


@both_profiles
def test_sparser_model_golden(blob_little):
    g = golden("digest_little.npz")
    m = capi.Model(blob_little)
    pcm = synth.batch_pcm((2, 31), 200, lead_silence=3)
    b = capi.Batch(m, 2)
    got = gpu_run(b, pcm)
    for i, s in enumerate((2, 31)):
        assert_bits_equal(got["gains"][:, i], g[f"s{s}_gains"], "gains")
        assert_bits_equal(got["vad"][:, i], g[f"s{s}_vad"], "vad")
        assert np.array_equal(crc_rows(got["out"][:, i]), g[f"s{s}_out_crc"])
        assert synth.crc32(b.export_state(i)) == int(g[f"s{s}_state_crc"])
    b.close()
    m.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65])
def test_multi_stream_invariance(model, blob_default, n):
    """stream k of a batch of N equals the same stream run alone: no cross-talk"""
    T = 12
    pcm = synth.batch_pcm([s % 7 for s in range(n)], T)
    b = capi.Batch(model, n)
    out, vad, gains = b.process(pcm)
    want = oracle_run(blob_default, pcm[:, :7] if n >= 7 else pcm)
    for s in range(n):
        r = s % 7 if n >= 7 else s
        assert_bits_equal(out[:, s], want["out"][:, r], f"pcm stream {s}")
        assert_bits_equal(gains[:, s], want["gains"][:, r], f"gains stream {s}")
        assert_bits_equal(vad[:, s], want["vad"][:, r], f"vad stream {s}")


def test_full_size_batch_properties(model, blob_default):
    """BASELINE config 2 size (4096 streams): size-independent checks -- replicated streams stay
    identical, and a sample of streams matches the oracle bit for bit."""
    N, T = 4096, 6
    base = synth.batch_pcm(range(16), T)
    pcm = np.ascontiguousarray(np.tile(base, (1, N // 16, 1)))
    b = capi.Batch(model, N)
    out, vad, gains = b.process(pcm)
    ref = out[:, :16]
    assert all(np.array_equal(out[:, k:k + 16].view(np.uint32), ref.view(np.uint32)) for k in range(0, N, 16))
    assert np.array_equal(gains.reshape(T, N // 16, 16, 32), np.broadcast_to(gains[:, None, :16], (T, N // 16, 16, 32)))
    want = oracle_run(blob_default, base)
    assert_bits_equal(out[:, :16], want["out"], "pcm")
    assert_bits_equal(gains[:, :16], want["gains"], "gains")
    assert_bits_equal(b.export_state(4095), want["state"][15], "state of the last stream")


def fuzz_pcm(n_streams, n_frames, seed):
    """signal shapes the fixed recipe of rnnoise_amd.synth does not reach: pure and two-tone periodic signals over
    the whole pitch range (period 30..800 samples, so that remove_doubling's sub-multiples and its minimum-period
    stop are exercised), period jumps, clicks, full-scale and near-silent levels, DC offsets, white and no noise"""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = n_frames * 480
    t = np.arange(n)
    out = np.zeros((n_frames, n_streams, 480), np.float32)
    for s_ in range(n_streams):
        period = rng.uniform(30, 800)
        if rng.random() < 0.3:  # period jump mid-way (continuity logic: prev_period / prev_gain)
            period = np.where(t < rng.integers(480, n), period, period * rng.choice([0.5, 2.0, 1.5, 0.98]))
        ph = 2 * np.pi * np.cumsum(1.0 / np.broadcast_to(period, (n,)))
        nh = int(rng.integers(1, 12))
        x = sum(rng.uniform(0.2, 1.0) * np.sin(k * ph + rng.uniform(0, 6.28)) for k in range(1, nh + 1))
        if rng.random() < 0.3:  # second, unrelated tone
            x = x + rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * t / rng.uniform(30, 800))
        x = x / max(np.abs(x).max(), 1e-9)
        amp = 10 ** rng.uniform(0.3, 4.5)  # 2 .. 31623: below the silence threshold up to clipping
        x = amp * x + rng.choice([0.0, 0.0, 30.0, 3000.0]) * rng.standard_normal(n) + rng.choice([0.0, 0.0, 500.0, -4000.0])
        if rng.random() < 0.2:
            x[rng.integers(0, n, 5)] += rng.choice([-30000.0, 30000.0], 5)  # clicks
        out[:, s_, :] = np.clip(np.rint(x), -32768, 32767).astype(np.float32).reshape(n_frames, 480)
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_signal_shapes_bit_exact(model, blob_default, seed):
    N, T = 160, 30
    pcm = fuzz_pcm(N, T, seed)
    b = capi.Batch(model, N)
    got = gpu_run(b, pcm)
    want = oracle_run(blob_default, pcm)
    assert len(np.unique(want["pitch"])) > 40, "the fuzz set must spread over the pitch range"
    assert_bits_equal(got["gains"], want["gains"], "gains")
    assert_bits_equal(got["vad"], want["vad"], "vad")
    assert_bits_equal(got["out"], want["out"], "pcm")
    f, sil, pit = b.debug_last()
    assert np.array_equal(pit, want["pitch"][-1]) and np.array_equal(sil, want["silence"][-1])
    assert_bits_equal(f, want["features"][-1], "features of the last frame")
    for s_ in range(0, N, 7):
        assert_bits_equal(b.export_state(s_), want["state"][s_], f"state of stream {s_}")


def test_16384_stream_batch_properties(model, blob_default):
    """16384 streams (4 network tiles per CU): replicated streams stay identical, a sample matches the
    oracle bit for bit, and one stream starts silent (network skipped, state frozen) while its tile
    neighbours run."""
    N, T = 16384, 6
    base = synth.batch_pcm(range(32), T)
    base[:3, 21] = 0                       # silent for 3 frames
    pcm = np.ascontiguousarray(np.tile(base, (1, N // 32, 1)))
    b = capi.Batch(model, N)
    out, vad, gains = b.process(pcm)
    for k in range(0, N, 32):
        assert np.array_equal(out[:, k:k + 32].view(np.uint32), out[:, :32].view(np.uint32)), f"tile at {k}"
        assert np.array_equal(gains[:, k:k + 32].view(np.uint32), gains[:, :32].view(np.uint32)), f"tile at {k}"
        assert np.array_equal(vad[:, k:k + 32].view(np.uint32), vad[:, :32].view(np.uint32)), f"tile at {k}"
    want = oracle_run(blob_default, base)
    assert want["silence"][:, 21].any()
    assert_bits_equal(out[:, :32], want["out"], "pcm")
    assert_bits_equal(gains[:, :32], want["gains"], "gains")
    assert_bits_equal(vad[:, :32], want["vad"], "vad")
    for s_ in (0, 15, 16, 21, 31):
        assert_bits_equal(b.export_state(N - 32 + s_), want["state"][s_], f"state of stream {N - 32 + s_}")


def test_state_export_import_round_trip_and_teacher_forcing(model, blob_default):
    pcm = synth.batch_pcm([5, 6], 40)
    b = capi.Batch(model, 2)
    b.process(pcm[:25])
    st = [b.export_state(0), b.export_state(1)]
    b2 = capi.Batch(model, 2)
    b2.import_state(0, st[1])  # swapped on purpose
    b2.import_state(1, st[0])
    o1, v1, g1 = b.process(pcm[25:])
    o2, v2, g2 = b2.process(pcm[25:, ::-1])
    assert_bits_equal(o1[:, 0], o2[:, 1], "pcm after import")
    assert_bits_equal(g1[:, 1], g2[:, 0], "gains after import")
    # an oracle state drives the GPU and vice versa
    o = Oracle(blob_default)
    o.run(pcm[:25, 0])
    assert_bits_equal(o.get_state(), st[0], "oracle vs exported state")
    bad = st[0].copy()
    bad[0] += 1.0  # analysis_mem no longer equals the tail of pitch_buf
    with pytest.raises(RuntimeError):
        b2.import_state(0, bad)


def test_reset_restores_initial_state(model):
    pcm = synth.batch_pcm([1, 2, 3], 10)
    b = capi.Batch(model, 3)
    a = b.process(pcm)
    b.reset()
    c = b.process(pcm)
    for x, y in zip(a, c):
        assert_bits_equal(x, y, "after reset")
    assert not a[0][0].any()  # first output frame is all zeros


@pytest.mark.parametrize("path", [0, 2])
def test_extreme_but_finite_inputs(model, blob_default, path):
    """the corners of the input domain the reference's own callers can reach: full-scale square waves, samples far outside
    the int16 range, a large DC offset, isolated impulses, amplitudes down in the denormal range (nothing flushes to zero:
    the x86 build runs without FTZ/DAZ), exact digital silence between bursts"""
    T = 30
    n = T * 480
    t = np.arange(n)
    rng = np.random.default_rng(5)
    sig = [
        32767.0 * np.sign(np.sin(2 * np.pi * 440 * t / 48000) + 1e-9),                 # full-scale square
        3.0e6 * np.sin(2 * np.pi * 233 * t / 48000),                                    # 100x beyond int16
        30000.0 + 2000.0 * rng.standard_normal(n),                                      # DC offset + noise
        np.where(t % 997 == 0, 32768.0, 0.0) - np.where(t % 1499 == 0, 32768.0, 0.0),   # impulses in silence
        1e-38 * rng.standard_normal(n),                                                 # denormal products all along the path
        np.where((t // 4800) % 2 == 0, 8000.0 * np.sin(2 * np.pi * 150 * t / 48000), 0.0),  # bursts / exact zeros
        1e-3 * rng.standard_normal(n),                                                  # just above the silence threshold
    ]
    ids = list(range(len(sig))) * 3                                                      # 21 streams: a full tile and a ragged one
    pcm = np.stack([sig[i].astype(np.float32).reshape(T, 480) for i in ids], axis=1)
    b = capi.Batch(model, len(ids))
    b.set_nn_path(path)
    out, vad, gains = b.process(pcm)
    assert np.isfinite(out).all()
    for i in range(len(sig)):
        want = Oracle(blob_default).run(pcm[:, i])
        for s in (i, i + len(sig), i + 2 * len(sig)):
            assert_bits_equal(out[:, s], want["out"], f"pcm, signal {i}, stream {s}")
            assert_bits_equal(gains[:, s], want["gains"], f"gains, signal {i}")
            assert_bits_equal(vad[:, s], want["vad"], f"vad, signal {i}")
    b.close()


@pytest.mark.parametrize("path", [1, 2])
def test_a_poisoned_stream_stays_alone(model, blob_default, path):
    """NaN / Inf samples in one stream of an MFMA tile (garbage in: its own output is not specified) must not reach its
    15 tile-mates or anybody else: streams share kernels, tiles and workgroups, never data"""
    T, n, bad = 12, 40, 21
    pcm = synth.batch_pcm([s % 9 for s in range(n)], T)
    pcm[3, bad, 100] = np.nan
    pcm[5, bad, 7] = np.inf
    pcm[6, bad, 300:320] = -np.inf
    b = capi.Batch(model, n)
    b.set_nn_path(path)
    out, vad, gains = b.process(pcm)
    cache = {}
    for s in range(n):
        if s == bad:
            continue
        if s % 9 not in cache:
            cache[s % 9] = Oracle(blob_default).run(pcm[:, s])
        assert_bits_equal(out[:, s], cache[s % 9]["out"], f"pcm of stream {s}")
        assert_bits_equal(gains[:, s], cache[s % 9]["gains"], f"gains of stream {s}")
    b.close()


def test_layerwise_network_survives_state_surgery(model, blob_default):
    """the layer-wise schedule keeps u8 images of the GRU state between frames (rn_dev.h: act_q); whatever else writes the
    state -- import, reset, a step on the other network kernels -- must invalidate them.  70 streams = ragged tile + ragged
    64-stream group."""
    n, T = 70, 24
    ids = [(5 * s) % 13 for s in range(n)]
    pcm = synth.batch_pcm(ids, T)
    pcm[9:12, 3::7] = 0                                   # silent gaps: a frozen state keeps its image, too
    want = {i: Oracle(blob_default).run(pcm[:, ids.index(i)]) for i in sorted(set(ids)) if ids.index(i) % 7 != 3}
    b = capi.Batch(model, n)
    b.set_nn_path(2)
    out1, _, g1 = b.process(pcm[:8])
    st = [b.export_state(s) for s in (0, 17, 69)]
    out2, _, g2 = b.process(pcm[8:16])
    # a second batch takes over three of the streams in the middle of the sequence
    b2 = capi.Batch(model, n)
    b2.set_nn_path(2)
    b2.process(pcm[:3])                                    # its own images are now valid for OTHER states
    for k, s in enumerate((0, 17, 69)):
        b2.import_state(s, st[k])
    o2b, _, g2b = b2.process(pcm[8:16])
    for s in (0, 17, 69):
        assert_bits_equal(o2b[:, s], out2[:, s], f"pcm of stream {s} after import")
        assert_bits_equal(g2b[:, s], g2[:, s], f"gains of stream {s} after import")
    # one step on the tile kernel in between, then layer-wise again
    b.set_nn_path(1)
    out3a, _, g3a = b.process(pcm[16:17])
    b.set_nn_path(2)
    out3b, _, g3b = b.process(pcm[17:])
    got_out = np.concatenate([out1, out2, out3a, out3b])
    got_g = np.concatenate([g1, g2, g3a, g3b])
    for s, i in enumerate(ids):
        if s % 7 != 3 and ids.index(i) == s:
            assert_bits_equal(got_out[:, s], want[i]["out"], f"pcm stream {s}")
            assert_bits_equal(got_g[:, s], want[i]["gains"], f"gains stream {s}")
    b.reset()
    o, _, g = b.process(pcm[:8])
    assert_bits_equal(o, out1, "after reset")
    assert_bits_equal(g, g1, "gains after reset")


def test_drop_in_single_stream_api(model, blob_default):
    """rnnoise_create / rnnoise_process_frame (in place, like examples/rnnoise_demo.c:57) / destroy"""
    T = 30
    pcm = synth.stream_pcm(42, T, lead_silence=3).astype(np.float32).reshape(T, 480)
    want = Oracle(blob_default).run(pcm)
    st1, st2 = capi.DenoiseState(model), capi.DenoiseState(model)
    other = synth.stream_pcm(43, T).astype(np.float32).reshape(T, 480)
    for t in range(T):
        y, vad = st1.process_frame(pcm[t])
        st2.process_frame(other[t])  # interleaved second state must not disturb the first
        assert_bits_equal(y, want["out"][t], f"frame {t}")
        assert np.float32(vad).view(np.uint32) == want["vad"][t].view(np.uint32)
    # caller-allocated storage (rnnoise_get_size + rnnoise_init, rnnoise.h:57,71)
    import ctypes as C
    L = capi.lib()
    buf = (C.c_char * L.rnnoise_get_size())()
    assert L.rnnoise_init(C.cast(buf, C.c_void_p), model.h) == 0
    x = pcm[0].copy()
    L.rnnoise_process_frame(C.cast(buf, C.c_void_p), x.ctypes.data_as(C.POINTER(C.c_float)),
                            x.ctypes.data_as(C.POINTER(C.c_float)))
    assert not x.any()


def test_device_resident_path_with_torch_stream(model, blob_default):
    """rnnoise_batch_process_device on torch-owned HBM and torch's current stream"""
    torch = pytest.importorskip("torch")
    N, T = 8, 5
    pcm = synth.batch_pcm(range(N), T)
    want = oracle_run(blob_default, pcm)
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(pcm).to(dev)
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty((T, N), device=dev)
    d_g = torch.empty((T, N, 32), device=dev)
    b = capi.Batch(model, N)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        b.process_device(d_out.data_ptr(), d_in.data_ptr(), d_vad.data_ptr(), d_g.data_ptr(), T,
                         torch.cuda.current_stream().cuda_stream)
    side.synchronize()
    assert_bits_equal(d_out.cpu().numpy(), want["out"], "pcm")
    assert_bits_equal(d_g.cpu().numpy(), want["gains"], "gains")
    assert_bits_equal(d_vad.cpu().numpy(), want["vad"], "vad")


def test_empty_calls_change_nothing(model, blob_default):
    """A call of ZERO frames (host and device entry points, float and int16) is a successful no-op -- state, later results and
    the caller's buffers untouched -- and a negative frame count fails without touching anything (the reference has no frame
    count: examples/rnnoise_demo.c:52-61 simply stops calling; the batched API's empty input is ours to define)."""
    torch = pytest.importorskip("torch")
    N, T = 5, 6
    pcm = synth.batch_pcm(range(N), T)
    want = oracle_run(blob_default, pcm)
    b = capi.Batch(model, N)
    empty = np.zeros((0, N, 480), np.float32)
    o, v, g = b.process(empty)
    assert o.shape == (0, N, 480) and v.shape == (0, N) and g.shape == (0, N, 32)
    a = b.process(pcm[:3])
    before = [b.export_state(s) for s in range(N)]
    b.process(empty)
    b.process_s16(np.zeros((0, N, 480), np.int16))
    dev = torch.device("cuda:0")
    d = torch.full((N, 480), 7.0, device=dev)
    b.process_device(d.data_ptr(), d.data_ptr(), 0, 0, 0)
    torch.cuda.synchronize()
    assert (d == 7.0).all()
    with pytest.raises(RuntimeError):
        b.process_device(d.data_ptr(), d.data_ptr(), 0, 0, -1)
    for s in range(N):
        assert_bits_equal(b.export_state(s), before[s], f"state of stream {s} across empty calls")
    c = b.process(pcm[3:])
    for k, name in enumerate(("out", "vad", "gains")):
        assert_bits_equal(np.concatenate([a[k], c[k]]), want[name], name)


# ---- batched MFMA network path (rnnoise_batch_set_nn_path(b, 1)) -------------------------------
@pytest.mark.parametrize("n,path", [(4, 1), (17, 1), (64, 1), (65, 1), (17, 0), (65, 0), (5, 2), (17, 2), (64, 2), (65, 2), (130, 2)])
def test_mfma_path_bit_exact(model, blob_default, n, path):
    """int8 MFMA on zero-filled dense tiles + f32 MFMA chains = same bits as the oracle, for tile
    counts that do and do not divide 16, with silent and non-silent streams mixed in one tile.
    path 2 = the layer-wise schedule large batches take (64 streams per GRU workgroup: ragged tiles AND ragged groups)"""
    T = 40
    ids = [(3 * s) % 11 for s in range(n)]
    pcm = synth.batch_pcm(ids, T, lead_silence=0)
    pcm[:5, ::3] = 0          # every third stream starts silent
    pcm[20:26, 1::4] = 0      # others go silent mid-way (state must freeze, src/denoise.c:474)
    uniq = sorted({(i, s % 3 == 0, s % 4 == 1) for s, i in enumerate(ids)})
    b = capi.Batch(model, n)
    import os
    one_max = int(os.environ.get("RNNOISE_AMD_NN_ONE_MAX", "512"))  # (test_throughput_kernels_at_small_sizes re-runs this with 0)
    assert b.set_nn_path(path) == (1 if n > one_max and n >= 16 else 0)   # documented default: up to 512 streams the latency-oriented vector kernel
    out, vad, gains = b.process(pcm)
    cache = {}
    for s, i in enumerate(ids):
        key = (i, s % 3 == 0, s % 4 == 1)
        if key not in cache:
            o = Oracle(blob_default)
            cache[key] = (o.run(pcm[:, s]), o.get_state())
        want, wstate = cache[key]
        assert_bits_equal(gains[:, s], want["gains"], f"gains stream {s}")
        assert_bits_equal(vad[:, s], want["vad"], f"vad stream {s}")
        assert_bits_equal(out[:, s], want["out"], f"pcm stream {s}")
        if s < 8 or s == n - 1:
            assert_bits_equal(b.export_state(s), wstate, f"state stream {s}")
    assert any(w["silence"].any() for w, _ in cache.values()) and len(uniq) >= 2


@pytest.mark.parametrize("n,path", [(70, 2), (37, 1)])
def test_sparser_blob_on_ragged_batches(blob_little, n, path):
    """BASELINE configs[3]'s blob (shorter, ragged block lists) on batch sizes with a partial tile and a partial group, layer-wise
    network and tile kernel (VERDICT r5 Weak #1: the sparser blob had only ever run on whole tiles)"""
    T = 24
    ids = [(5 * s) % 7 for s in range(n)]
    pcm = synth.batch_pcm(ids, T, lead_silence=0)
    pcm[:4, 2::5] = 0
    m = capi.Model(blob_little)
    b = capi.Batch(m, n)
    b.set_nn_path(path)
    out, vad, gains = b.process(pcm)
    cache = {}
    for s, i in enumerate(ids):
        key = (i, s % 5 == 2)
        if key not in cache:
            cache[key] = Oracle(blob_little).run(pcm[:, s])
        want = cache[key]
        assert_bits_equal(gains[:, s], want["gains"], f"gains stream {s}")
        assert_bits_equal(vad[:, s], want["vad"], f"vad stream {s}")
        assert_bits_equal(out[:, s], want["out"], f"pcm stream {s}")
    b.close()
    m.close()


@both_profiles
def test_mfma_and_vector_paths_agree_on_golden(model):
    g = golden("digest_default.npz")
    streams = (0, 1, 159, 4095)
    pcm = synth.batch_pcm(streams, 400, lead_silence=5)
    b = capi.Batch(model, 4)
    assert b.set_nn_path(1) == 0          # small batches default to the vector path (covered by the digest test above)
    got = gpu_run(b, pcm)
    for i, s in enumerate(streams):
        assert_bits_equal(got["gains"][:, i], g[f"s{s}_gains"], "gains")
        assert_bits_equal(got["vad"][:, i], g[f"s{s}_vad"], "vad")
        assert np.array_equal(crc_rows(got["out"][:, i]), g[f"s{s}_out_crc"])
        assert synth.crc32(b.export_state(i)) == int(g[f"s{s}_state_crc"])


@both_profiles
def test_mfma_sparser_model(blob_little):
    g = golden("digest_little.npz")
    m = capi.Model(blob_little)
    pcm = synth.batch_pcm((2, 31), 200, lead_silence=3)
    b = capi.Batch(m, 2)
    b.set_nn_path(1)
    got = gpu_run(b, pcm)
    for i, s in enumerate((2, 31)):
        assert_bits_equal(got["gains"][:, i], g[f"s{s}_gains"], "gains")
        assert np.array_equal(crc_rows(got["out"][:, i]), g[f"s{s}_out_crc"])
    b.close()
    m.close()


# ---- software pipeline bookkeeping (3 streams, 6-slot pitch ring, 3 spectra slots) ------------------
def test_pipeline_chunking_is_invisible(model, blob_default):
    """the same 45 frames fed as calls of 1, 2, 5, 1, 3, 7, ... frames (multi-frame calls are pipelined over
    three HIP streams, single-frame calls are not) give the oracle's bits, whatever the call boundaries"""
    streams = [2, 9, 33, 64, 101]
    T = 45
    pcm = synth.batch_pcm(streams, T, lead_silence=1)
    pcm[17:20, 1] = 0  # a silent gap in one stream (network state must freeze across a call boundary too)
    want = oracle_run(blob_default, pcm)
    for path in (2, 1, 0, -1):     # -1: a different network schedule at every call (they share all state)
        b = capi.Batch(model, len(streams))
        if path >= 0:
            b.set_nn_path(path)
        outs, vads, gains = [], [], []
        t = 0
        for n in [1, 2, 5, 1, 3, 7, 1, 1, 4, 6, 2, 12]:
            if path < 0:
                b.set_nn_path((2, 0, 1)[len(outs) % 3])
            o, v, g = b.process(pcm[t:t + n])
            outs.append(o); vads.append(v); gains.append(g)
            t += n
        assert t == T
        assert_bits_equal(np.concatenate(outs), want["out"], f"pcm (nn path {path})")
        assert_bits_equal(np.concatenate(gains), want["gains"], "gains")
        assert_bits_equal(np.concatenate(vads), want["vad"], "vad")
        for i in range(len(streams)):
            assert_bits_equal(b.export_state(i), want["state"][i], f"state {i}")


@pytest.mark.parametrize("mode", ["1", "9"])
def test_other_stream_schedules_forced(mode):
    """RNNOISE_AMD_PIPE (read once per process) selects the A/B schedules -- 1: only K0 on a side stream, 9: no side
    streams; the chunking / in-place / MFMA parity cases must not notice"""
    import os
    import subprocess
    import sys
    if os.environ.get("RNNOISE_AMD_PIPE"):
        pytest.skip("already inside a forced run")
    env = dict(os.environ, RNNOISE_AMD_PIPE=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k",
                        "test_pipeline_chunking or test_device_call_in_place or test_mfma_path_bit_exact"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_throughput_kernels_at_small_sizes():
    """Small batches run the latency-oriented kernels by default (rn_hp_one_kernel up to 2048 streams, rn_nn_one_kernel up to
    512 on the vector path); with both switched off ($RNNOISE_AMD_HP_ONE_MAX / $RNNOISE_AMD_NN_ONE_MAX = 0, read once per
    process) the same cases go through rn_hp_kernel and rn_nn_vector_kernel -- the kernels of larger batches -- and must give
    the same bits; $RNNOISE_AMD_K1_SPW=4 adds the four-stream analysis workgroups of large batches, tails included"""
    import os
    import subprocess
    import sys
    if os.environ.get("RNNOISE_AMD_NN_ONE_MAX"):
        pytest.skip("already inside a forced run")
    # ... and with four streams per analysis workgroup (rn_analysis_kernel, the form of batches from 2560 streams up): the batch
    # sizes of these cases are not multiples of four, so the tail workgroup's surplus waves -- which redo the last stream, meet
    # every barrier and lend their arenas to the narrow phases' row and pair passes -- are exercised too
    # ... and the tile network kernel in its sixteen-wave form (rn_nn_mfma16_kernel: by default only one-frame calls on up to 4,096
    # streams take it; the multi-frame calls of these cases run the eight-wave form in the default run of the suite)
    env = dict(os.environ, RNNOISE_AMD_NN_ONE_MAX="0", RNNOISE_AMD_HP_ONE_MAX="0", RNNOISE_AMD_K1_SPW="4", RNNOISE_AMD_TILE_WAVES="16")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, os.path.join(root, "tests", "test_blob_tools.py"), "-k",
                        "test_mfma_path_bit_exact or test_synthetic_models_on_gpu or test_s16_entry_points or test_drop_in_single_stream_api"],
                       env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_at_size_kernels_on_small_ragged_batches():
    """The kernels that only LARGE batches take by default -- the four-wave GRU layer kernel (rn_nn_gru_kernel: more 64-stream
    groups than CUs), the lane-per-stream high-pass (rn_hp_kernel: above 2,048 streams), the four-stream analysis workgroup -- forced
    onto the small ragged cases ($RNNOISE_AMD_GRU_VARIANT=w4, with the latency kernels off and the layer-wise network from size 0
    up): n = 4 ... 130 streams against the oracle stream by stream, partial tiles, partial groups, partial waves.  The default run of
    the same cases takes the eight-wave form (w8) and the wave-per-stream high-pass; tests/test_gpu_at_size.py has the ragged batch at
    size (40,037 streams)."""
    import os
    import subprocess
    import sys
    if os.environ.get("RNNOISE_AMD_NN_ONE_MAX"):
        pytest.skip("already inside a forced run")
    env = dict(os.environ, RNNOISE_AMD_NN_ONE_MAX="0", RNNOISE_AMD_HP_ONE_MAX="0", RNNOISE_AMD_K1_SPW="4", RNNOISE_AMD_NN_LAYERS_MIN="0",
               RNNOISE_AMD_GRU_VARIANT="w4")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, os.path.join(root, "tests", "test_blob_tools.py"), "-k",
                        "test_mfma_path_bit_exact or test_sparser_blob_on_ragged_batches or test_synthetic_models_on_gpu or test_s16_entry_points"],
                       env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_device_call_in_place(model, blob_default):
    """d_out == d_in (the reference demo processes in place, examples/rnnoise_demo.c:57)"""
    torch = pytest.importorskip("torch")
    N, T = 5, 9
    pcm = synth.batch_pcm(range(N), T)
    want = oracle_run(blob_default, pcm)
    buf = torch.from_numpy(pcm).cuda()
    b = capi.Batch(model, N)
    b.set_nn_path(1)
    b.process_device(buf.data_ptr(), buf.data_ptr(), 0, 0, T, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert_bits_equal(buf.cpu().numpy(), want["out"], "pcm in place")


def x86_float_to_short(x):
    """`short s = f` as x86-64 compiles it (what examples/rnnoise_demo.c:58 does to every output sample): cvttss2si to 32 bits --
    truncation toward zero, 0x80000000 for NaN or out of range -- then the low 16 bits"""
    x = np.asarray(x, np.float32)
    ok = (x >= np.float32(-2147483648.0)) & (x < np.float32(2147483648.0))
    i = np.where(ok, np.trunc(np.where(ok, x, 0)), -2147483648.0).astype(np.int64)
    return (i & 0xFFFF).astype(np.uint16).view(np.int16)


@pytest.mark.parametrize("n", [5, 70])
def test_s16_entry_points(model, blob_default, n):
    """rnnoise_batch_process_s16 / _device_s16: int16 PCM in and out, converted inside the first and the last kernel of the step
    as the reference's only caller converts around its call (examples/rnnoise_demo.c:56,58).  Bits = the float path's,
    then that cast; VAD, gains and the exported state are the float path's (the oracle's); float and s16 calls mix freely."""
    torch = pytest.importorskip("torch")
    T = 21
    pcm = synth.batch_pcm(range(100, 100 + n), T, lead_silence=1)      # s16-valued floats
    want = oracle_run(blob_default, pcm)
    pcm16 = pcm.astype(np.int16)
    assert np.array_equal(pcm16.astype(np.float32), pcm)
    b = capi.Batch(model, n)
    o1, v1, g1 = b.process_s16(pcm16[:8])                              # host-fed, multi-frame (pipelined)
    o2, v2, g2 = b.process(pcm[8:9])                                   # a float call in between
    d_in = torch.from_numpy(pcm16[9:]).cuda()
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty((T - 9, n), device="cuda")
    d_g = torch.empty((T - 9, n, 32), device="cuda")
    b.process_device(d_out.data_ptr(), d_in.data_ptr(), d_vad.data_ptr(), d_g.data_ptr(), T - 9,
                     torch.cuda.current_stream().cuda_stream, s16=True)
    torch.cuda.synchronize()
    got16 = np.concatenate([o1, x86_float_to_short(o2), d_out.cpu().numpy()])
    assert o1.dtype == np.int16 and np.array_equal(got16, x86_float_to_short(want["out"]))
    assert_bits_equal(o2, want["out"][8:9], "the float call between the s16 calls")
    assert_bits_equal(np.concatenate([v1, v2, d_vad.cpu().numpy()]), want["vad"], "vad")
    assert_bits_equal(np.concatenate([g1, g2, d_g.cpu().numpy()]), want["gains"], "gains")
    for i in (0, n - 1):
        assert_bits_equal(b.export_state(i), want["state"][i], f"state {i}")
    b.close()


def test_s16_out_of_range_samples_wrap_like_the_x86_cast(model, blob_default):
    """the demo's float -> short cast is not a saturating one: an output sample beyond +-32767 keeps the low 16 bits of its
    32-bit truncation (and 0 once it leaves the int32 range).  Input 100x full scale as floats drives the output there."""
    N, T = 4, 6
    pcm = synth.batch_pcm(range(N), T) * np.float32(300.0)
    pcm[:, 3] *= np.float32(1e6)                                        # beyond the int32 range
    want = oracle_run(blob_default, pcm)
    assert np.abs(want["out"]).max() > 2.2e9 and (np.abs(want["out"][:, :3]) > 40000).any()
    b = capi.Batch(model, N)
    # the float input cannot go through the s16 door; feed floats, take s16 out on the device path
    torch = pytest.importorskip("torch")
    d_in = torch.from_numpy(pcm).cuda()
    out_f, _, _ = capi.Batch(model, N).process(pcm)
    assert_bits_equal(out_f, want["out"], "float path on out-of-range input")
    # mixed door: float in / s16 out does not exist in the API, so check the cast on the synthesis side through a state copy:
    # run T-1 frames as float, then one s16 frame whose INPUT is in range but whose synth_mem overlap is far out of range
    b.process(pcm[:T - 1])
    last16 = np.clip(pcm[T - 1:], -32768, 32767).astype(np.int16)
    o16, _, _ = b.process_s16(last16)
    ref = capi.Batch(model, N)
    ref.process(pcm[:T - 1])
    of, _, _ = ref.process(last16.astype(np.float32))
    assert (np.abs(of) > 40000).any()
    assert np.array_equal(o16, x86_float_to_short(of))
    del d_in


# ---- rcpps profiles (include/rnnoise_amd.h: rnnoise_amd_set_rcp_profile; reference: src/vec_avx.h:413,442,484,505) -------
@pytest.mark.parametrize("profile", [pytest.param("amd-zen5", marks=pytest.mark.rcp("amd-zen5")),
                                     pytest.param("host", marks=pytest.mark.rcp("host"))])
def test_every_network_kernel_follows_the_rcp_profile(model, blob_default, profile):
    """the three network schedules (vector, 16-stream MFMA tile, layer-wise) and the oracle on the SAME non-default profile:
    identical bits -- and different from the Intel profile's, so the table really is what the kernels read"""
    assert capi.rcp_profile().startswith(profile)
    T, n = 24, 70
    pcm = synth.batch_pcm([(5 * s) % 13 for s in range(n)], T, lead_silence=1)
    want = {}
    for path in (0, 1, 2):
        b = capi.Batch(model, n)
        b.set_nn_path(path)
        out, vad, gains = b.process(pcm)
        for s in (0, 1, 17, 64, 69):
            if s not in want:
                want[s] = Oracle(blob_default).run(pcm[:, s])
            assert_bits_equal(gains[:, s], want[s]["gains"], f"{profile} path {path} gains stream {s}")
            assert_bits_equal(vad[:, s], want[s]["vad"], f"{profile} path {path} vad stream {s}")
            assert_bits_equal(out[:, s], want[s]["out"], f"{profile} path {path} pcm stream {s}")
        b.close()
    if not capi.rcp_profile().endswith("intel"):
        from oracle import binding
        binding.set_rcp_profile("intel")
        other = Oracle(blob_default).run(pcm[:, 0])
        assert not np.array_equal(other["gains"], want[0]["gains"])


def test_profile_switch_reaches_live_batches(model, blob_default):
    """rnnoise_amd_set_rcp_profile() replaces the table of every device that already holds one: a batch created before the
    switch computes the next frames on the new profile (state carried over) -- checked against an oracle switched the same way"""
    from oracle import binding
    T = 10
    pcm = synth.batch_pcm([2, 9, 4], 2 * T, lead_silence=0)
    b = capi.Batch(model, 3)
    o = [Oracle(blob_default) for _ in range(3)]
    got1 = b.process(pcm[:T])
    w1 = [o[s].run(pcm[:T, s]) for s in range(3)]
    capi.set_rcp_profile("amd-zen5")
    binding.set_rcp_profile("amd-zen5")
    got2 = b.process(pcm[T:])
    w2 = [o[s].run(pcm[T:, s]) for s in range(3)]
    for s in range(3):
        assert_bits_equal(got1[0][:, s], w1[s]["out"], f"before the switch, stream {s}")
        assert_bits_equal(got2[0][:, s], w2[s]["out"], f"after the switch, stream {s}")
        assert_bits_equal(got2[2][:, s], w2[s]["gains"], f"after the switch, gains {s}")
    assert capi.rcp_profile() == "amd-zen5"
    with pytest.raises(ValueError):
        capi.set_rcp_profile("vax")
