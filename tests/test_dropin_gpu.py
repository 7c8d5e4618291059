"""GPU suite: the rnnoise.h drop-in boundary (reference include/rnnoise.h:51-125) on device-resident pooled states,
self-contained caller-memory states, the NULL-model fallback, concurrent states on several threads, and the host-fed
batched path (pinned / pageable memory, several chunks in flight)."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal
from oracle.binding import Oracle
from rnnoise_amd import capi, synth
from test_gpu_parity import oracle_run

pytestmark = pytest.mark.gpu
FP = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def model(blob_default):
    return capi.Model(blob_default)


@pytest.mark.parametrize("n_threads", [4, 32])
def test_pooled_states_on_four_threads(model, blob_default, n_threads):
    """70 rnnoise_create()d states (more than a launch group's 64 entries), driven from 4 and from 32 threads at once: every stream gets the
    oracle's bits -- states are independent, no global lock serialises them into one another's data, and the combiner that
    gathers concurrent calls into shared launches (dropin.cpp) keeps every row at its own frame phase: state s starts s % 4
    frames late, so the rows of one launch group sit at different ring and spectra slots"""
    T, n = 12, 70
    pcm = [synth.stream_pcm(s % 9, T, lead_silence=s % 3).astype(np.float32).reshape(T, 480) for s in range(n)]
    want = {}
    for s in range(n):
        if (s % 9, s % 3) not in want:
            want[(s % 9, s % 3)] = Oracle(blob_default).run(pcm[s])
    states = [capi.DenoiseState(model) for _ in range(n)]
    got_out = [np.zeros((T, 480), np.float32) for _ in range(n)]
    got_vad = [np.zeros(T, np.float32) for _ in range(n)]
    errs = []

    def work(tid):
        try:
            for t in range(T + 3):
                for s in range(tid, n, n_threads):
                    f = t - s % 4
                    if 0 <= f < T:
                        y, v = states[s].process_frame(pcm[s][f])
                        got_out[s][f], got_vad[s][f] = y, v
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for s in range(n):
        ref = want[(s % 9, s % 3)]
        assert_bits_equal(got_out[s], ref["out"], f"pcm of state {s}")
        assert_bits_equal(got_vad[s], ref["vad"], f"vad of state {s}")
    for st in states:
        st.close()
    # rows are recycled zeroed: a new state starts from the rnnoise_init() state again
    st = capi.DenoiseState(model)
    y, v = st.process_frame(pcm[3][0])
    assert_bits_equal(y, Oracle(blob_default).run(pcm[3][:1])["out"][0], "first frame of a recycled row")


def test_three_hundred_states_from_ninety_six_threads(model, blob_default):
    """more states than a pool has rows (256 by default: the 257th opens a second pool, on the next device if there is one) and
    more concurrent callers than a launch group has entries (64): the combiner launches the oldest 64 queued requests and leaves
    the rest for the next group; rows sit at staggered frame phases; every stream gets the oracle's bits"""
    T, n, n_threads = 6, 300, 96
    pcm = [synth.stream_pcm(s % 7, T, lead_silence=s % 2).astype(np.float32).reshape(T, 480) for s in range(n)]
    want = {}
    for s in range(n):
        if (s % 7, s % 2) not in want:
            want[(s % 7, s % 2)] = Oracle(blob_default).run(pcm[s])
    states = [capi.DenoiseState(model) for _ in range(n)]
    got_out = [np.zeros((T, 480), np.float32) for _ in range(n)]
    got_vad = [np.zeros(T, np.float32) for _ in range(n)]
    errs = []

    def work(tid):
        try:
            for t in range(T + 2):
                for s in range(tid, n, n_threads):
                    f = t - s % 3
                    if 0 <= f < T:
                        got_out[s][f], got_vad[s][f] = states[s].process_frame(pcm[s][f])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for s in range(n):
        ref = want[(s % 7, s % 2)]
        assert_bits_equal(got_out[s], ref["out"], f"pcm of state {s}")
        assert_bits_equal(got_vad[s], ref["vad"], f"vad of state {s}")
    for st in states:
        st.close()


def test_batch_and_pooled_state_on_the_last_device(blob_default):
    """a node with several GPUs: a batch created on the LAST device, and a process whose rnnoise_create() states are sent there
    ($RNNOISE_AMD_DEVICE), give the oracle's bits -- device 0 is not special.  (One GPU: skipped; the 8-GPU run is the driver's.)"""
    n_dev = capi.lib().rnnoise_amd_device_count()
    if n_dev < 2:
        pytest.skip("one GPU visible")
    T = 6
    pcm = synth.batch_pcm(range(40), T, lead_silence=1)
    want = oracle_run(blob_default, pcm, collect_state=False)
    m = capi.Model(blob_default)
    b = capi.Batch(m, 40, device=n_dev - 1)
    out, vad, gains = b.process(pcm)
    assert_bits_equal(out, want["out"], "pcm on the last device")
    assert_bits_equal(gains, want["gains"], "gains on the last device")
    b.close()
    code = r"""
import sys, zlib, lzma, numpy as np
sys.path.insert(0, %r)
from rnnoise_amd import capi, synth
m = capi.Model(lzma.decompress(open(%r, "rb").read()))
st = [capi.DenoiseState(m) for _ in range(3)]
x = synth.batch_pcm(range(3), 6, lead_silence=1)
y = np.stack([np.stack([st[s].process_frame(x[t, s])[0] for s in range(3)]) for t in range(6)])
print("CRC", zlib.crc32(y.tobytes()))
""" % (ROOT, os.path.join(ROOT, "tests", "golden", "default.blob.xz"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       env=dict(os.environ, RNNOISE_AMD_DEVICE=str(n_dev - 1)), timeout=300)
    import zlib
    assert r.returncode == 0 and f"CRC {zlib.crc32(np.ascontiguousarray(want['out'][:, :3]).tobytes())}" in r.stdout, r.stdout + r.stderr


def test_a_failed_launch_group_restarts_its_states_from_zero(blob_default, tmp_path):
    """(fault injection: $RNNOISE_AMD_TEST_FAIL_GROUP exists in the instrumented library only -- the product reads no such switch)
    a launch group that fails after its high-pass has run leaves the row's pitch ring one frame ahead of the host-side slot
    counters.  The frame comes back zeroed (VAD 0), and from the next call on the state is a FRESH one (what rnnoise_init leaves):
    its output equals the oracle started at that frame -- not a stream running one ring slot off"""
    code = r"""
import sys, lzma, numpy as np
sys.path.insert(0, %r)
from rnnoise_amd import capi, synth
m = capi.Model(lzma.decompress(open(%r, "rb").read()))
st = capi.DenoiseState(m)
x = synth.stream_pcm(5, 9).astype(np.float32).reshape(9, 480)
out = np.stack([st.process_frame(x[t])[0] for t in range(9)])
np.save(%r, out)
"""
    blob_path = os.path.join(ROOT, "tests", "golden", "default.blob.xz")
    out_path = str(tmp_path / "out.npy")
    # (group 0 is frame 0, ...: the 4th group of the process is this state's frame 3)
    r = subprocess.run([sys.executable, "-c", code % (ROOT, blob_path, out_path)], capture_output=True, text=True,
                       env=dict(os.environ, RNNOISE_AMD_TEST_FAIL_GROUP="3", RNNOISE_AMD_LIB=capi.INSTR_LIB_PATH), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "returning a zeroed frame" in r.stderr and "restarts from zero" in r.stderr, r.stderr
    got = np.load(out_path)
    x = synth.stream_pcm(5, 9).astype(np.float32).reshape(9, 480)
    before = Oracle(blob_default).run(x[:3])
    assert_bits_equal(got[:3], before["out"], "frames before the failure")
    assert not got[3].any()
    after = Oracle(blob_default).run(x[4:])       # a fresh state fed frames 4 ..
    assert_bits_equal(got[4:], after["out"], "frames after the restart")


def test_failed_launch_groups_strand_nobody(tmp_path):
    """a pool holds up to 1,024 rows but a launch group takes the oldest 64 queued requests, so requests can
    still be QUEUED behind a group whose launch fails -- and with no other group in flight nobody adopted them: their callers slept
    for ever.  160 states on 160 threads marching in step (a barrier per frame), seven launch groups of the process made to fail:
    every call must come back (the subprocess has a timeout), failed frames zeroed, and the last frames -- by then every poisoned
    state has restarted -- must be somebody's output again."""
    code = r"""
import sys, lzma, threading, numpy as np
sys.path.insert(0, %r)
from rnnoise_amd import capi, synth
m = capi.Model(lzma.decompress(open(%r, "rb").read()))
N, T = 160, 8
st = [capi.DenoiseState(m) for _ in range(N)]
x = synth.stream_pcm(3, T).astype(np.float32).reshape(T, 480)
bar = threading.Barrier(N)
nz = [0] * N
def run(k):
    for t in range(T):
        bar.wait()
        out, vad = st[k].process_frame(x[t])
        if t == T - 1: nz[k] = int(np.any(out != 0))
th = [threading.Thread(target=run, args=(k,)) for k in range(N)]
[t.start() for t in th]; [t.join() for t in th]
print("DONE", sum(nz))
"""
    blob_path = os.path.join(ROOT, "tests", "golden", "default.blob.xz")
    r = subprocess.run([sys.executable, "-c", code % (ROOT, blob_path)], capture_output=True, text=True,
                       env=dict(os.environ, RNNOISE_AMD_TEST_FAIL_GROUP="2,3,4,5,6,7,8", RNNOISE_AMD_LIB=capi.INSTR_LIB_PATH), timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "returning a zeroed frame" in r.stderr, r.stderr[-2000:]
    assert "DONE 160" in r.stdout, r.stdout[-500:]   # frame 7: every state -- restarted or not -- produces sound again


def test_caller_memory_states_interleaved(model, blob_default):
    """rnnoise_get_size() + rnnoise_init() on caller memory (rnnoise.h:57,71): self-contained POD, may be copied"""
    L = capi.lib()
    T = 10
    pcm = [synth.stream_pcm(20 + k, T, lead_silence=1).astype(np.float32).reshape(T, 480) for k in range(2)]
    want = [Oracle(blob_default).run(p) for p in pcm]
    size = L.rnnoise_get_size()
    bufs = [(C.c_char * size)() for _ in range(2)]
    for b in bufs:
        assert L.rnnoise_init(C.cast(b, C.c_void_p), model.h) == 0
    for t in range(T):
        if t == 5:  # a state in caller memory is plain data: moving it must not matter
            moved = (C.c_char * size)()
            C.memmove(moved, bufs[0], size)
            bufs[0] = moved
        for k in range(2):
            x = pcm[k][t].copy()
            L.rnnoise_process_frame.restype = C.c_float
            v = L.rnnoise_process_frame(C.cast(bufs[k], C.c_void_p), x.ctypes.data_as(FP), x.ctypes.data_as(FP))
            assert_bits_equal(x, want[k]["out"][t], f"state {k} frame {t}")
            assert np.float32(v).view(np.uint32) == want[k]["vad"][t].view(np.uint32)


def test_null_model_uses_the_default_blob(tmp_path, blob_default):
    """model == NULL (rnnoise.h:64-76): $RNNOISE_AMD_DEFAULT_MODEL stands in for the compiled-in weights; without it the
    call fails cleanly (NULL / -1) instead of crashing"""
    blob_path = tmp_path / "weights_blob.bin"
    blob_path.write_bytes(blob_default)
    code = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r)
from rnnoise_amd import capi, synth
L = capi.lib()
st = L.rnnoise_create(None)
print("CREATE", bool(st))
if st:
    x = synth.stream_pcm(4, 3).astype(np.float32).reshape(3, 480)
    out = []
    for t in range(3):
        y = x[t].copy()
        L.rnnoise_process_frame(st, capi._fp(y), capi._fp(y))
        out.append(y)
    print("CRC", __import__("zlib").crc32(np.stack(out).tobytes()))
    L.rnnoise_destroy(st)
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       env=dict(os.environ, RNNOISE_AMD_DEFAULT_MODEL=str(blob_path)), timeout=300)
    assert r.returncode == 0 and "CREATE True" in r.stdout, r.stdout + r.stderr
    import zlib
    x = synth.stream_pcm(4, 3).astype(np.float32).reshape(3, 480)
    assert f"CRC {zlib.crc32(Oracle(blob_default).run(x)['out'].tobytes())}" in r.stdout
    env = {k: v for k, v in os.environ.items() if k != "RNNOISE_AMD_DEFAULT_MODEL"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "CREATE False" in r.stdout and "no default weight blob" in r.stderr, r.stdout + r.stderr


def test_bad_state_returns_a_zeroed_frame_instead_of_aborting():
    L = capi.lib()
    junk = (C.c_char * L.rnnoise_get_size())()
    x = np.ones(480, np.float32)
    L.rnnoise_process_frame.restype = C.c_float
    v = L.rnnoise_process_frame(C.cast(junk, C.c_void_p), x.ctypes.data_as(FP), x.ctypes.data_as(FP))
    assert v == 0.0 and not x.any()


def test_c_threads_through_the_combiner_give_the_oracles_bits(blob_default, tmp_path):
    """tools/configs0_mt.c in check mode -- plain C threads, no interpreter lock in the way: 64 threads over 300 rnnoise_create()d
    states, 200 frames each; state k is fed the deterministic signal k % 7 and checksums every output sample and VAD value.  All
    states of a signal agree, and the seven checksums are the ORACLE's for the same signals: no frame lost, repeated or delivered
    to the wrong row by the combiner.  (1,000 frames, up to 1,024 states and 128 threads: profiles/r5_combiner_stress.txt)"""
    exe, blob_path = str(tmp_path / "configs0_mt"), str(tmp_path / "default.blob")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "configs0_mt.c"), "-o", exe,
                    "-L" + os.path.join(ROOT, "rnnoise_amd"), "-l:librnnoise_amd.so", "-Wl,-rpath," + os.path.join(ROOT, "rnnoise_amd"), "-lpthread"],
                   check=True, capture_output=True, text=True)
    with open(blob_path, "wb") as f:
        f.write(blob_default)
    frames = 200
    r = subprocess.run([exe, blob_path, "64", str(frames - 100), "300", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "every state equal to its siblings" in r.stdout, r.stdout + r.stderr
    got = r.stdout.split("checksums")[1].split()
    M32, M64 = 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF
    want = []
    for k in range(7):
        g = (977 * k + 1) & M32
        x = np.empty((frames, 480), np.float32)
        for t in range(frames):
            for i in range(480):
                g = (g * 1664525 + 1013904223) & M32
                x[t, i] = float((g >> 18) - 8192)
        ref = Oracle(blob_default).run(x)
        h = 0xcbf29ce484222325
        out, vad = ref["out"].view(np.uint32), ref["vad"].view(np.uint32)
        for t in range(frames):
            for u in out[t].tolist():
                h = ((h ^ u) * 0x100000001b3) & M64
            h = ((h ^ int(vad[t])) * 0x100000001b3) & M64
        want.append(f"{h:016x}")
    assert got == want


def test_host_fed_soak_on_the_named_copy_engines():
    """tools/soak_hostio.py, short: 10 calls of 16 int16 frames at 65,536 streams and 10 calls of 12 float frames at 32,768 from pinned
    memory on the default copy mode (uploads and downloads on two named copy engines, ordered against the kernels by count words and
    release kernels) -- every ring slot reused ~25 times; after every call all replicas equal and the first block bit-identical to
    the oracle.  (300 calls each: profiles/r5_soak_hostio.txt)"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("RNNOISE_AMD_HOSTIO_COPY", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_hostio.py"), "10"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.count("(0 bad calls") == 2, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("pinned", [False, True, "s16"])
def test_host_fed_path_chunks_in_flight(model, blob_default, pinned):
    """rnnoise_batch_process: 8192 streams x 12 frames from pageable memory (6 chunks of 32 MB through the double-buffered
    bounce buffers) and from pinned memory (direct DMA frame by frame through the 6-slot ring in HBM, beside ONE pipelined
    12-frame device call: every slot is reused once), the latter also with int16 PCM; replicas stay identical and match the
    oracle; a pageable call then continues the same streams"""
    torch = pytest.importorskip("torch")
    N, T = 8192, 12
    base = synth.batch_pcm(range(8), T, lead_silence=1)
    pcm = np.ascontiguousarray(np.tile(base, (1, N // 8, 1)))
    b = capi.Batch(model, N)
    if pinned == "s16":
        t_in = torch.from_numpy(pcm.astype(np.int16)).pin_memory()
        t_out = torch.empty_like(t_in).pin_memory()
        t_vad = torch.empty((T, N)).pin_memory()
        t_g = torch.empty((T, N, 32)).pin_memory()
        b.process_into(t_out.data_ptr(), t_in.data_ptr(), t_vad.data_ptr(), t_g.data_ptr(), T, s16=True)
        out16, vad, gains = t_out.numpy(), t_vad.numpy(), t_g.numpy()
        want = oracle_run(blob_default, base, collect_state=False)
        from test_gpu_parity import x86_float_to_short
        assert np.array_equal(out16[:, :8], x86_float_to_short(want["out"])) and np.array_equal(out16[:, -8:], out16[:, :8])
        out = np.ascontiguousarray(np.tile(want["out"], (1, N // 8, 1)))  # (PCM checked above; the rest below as for floats)
    elif pinned:
        t_in = torch.from_numpy(pcm).pin_memory()
        t_out = torch.empty_like(t_in).pin_memory()
        t_vad = torch.empty((T, N)).pin_memory()
        t_g = torch.empty((T, N, 32)).pin_memory()
        assert capi.lib().rnnoise_batch_process(b.h, C.cast(t_out.data_ptr(), FP), C.cast(t_in.data_ptr(), FP),
                                                C.cast(t_vad.data_ptr(), FP), C.cast(t_g.data_ptr(), FP), T) == 0
        out, vad, gains = t_out.numpy(), t_vad.numpy(), t_g.numpy()
    else:
        out, vad, gains = b.process(pcm)
    o4 = out.reshape(T, N // 8, 8, 480).view(np.uint32)
    assert (o4 == o4[:, :1]).all()
    want = oracle_run(blob_default, base, collect_state=False)
    assert_bits_equal(out[:, :8], want["out"], "pcm")
    assert_bits_equal(out[:, -8:], want["out"], "pcm of the last block")
    assert_bits_equal(vad[:, :8], want["vad"], "vad")
    assert_bits_equal(gains[:, 8:16], want["gains"], "gains")
    # a second call continues the streams (state carried across calls and chunk boundaries)
    out2, _, _ = b.process(pcm[:2])
    o = Oracle(blob_default)
    ref2 = o.run(np.concatenate([base[:, 3], base[:2, 3]]))["out"][T:]
    assert_bits_equal(out2[:, 3], ref2, "continuation")
