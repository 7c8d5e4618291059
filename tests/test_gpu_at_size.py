"""GPU suite, at BASELINE.json's full sizes: configs[2] (65,536 streams, the bench default; runs the >= 24,576-stream
build of the analysis kernel and 4096 MFMA tiles) and configs[3] (sparser blob, 32,768 streams).  An oracle run of
65,536 streams would take hours, so the checks are the size-independent ones: tiled replicas of a 32-stream block must
stay bit-identical to each other (no cross-talk, no dependence on the tile / CU / XCD a stream lands on), and the block
itself is checked bit for bit against the oracle, including a stream that starts silent inside a live MFMA tile.
Also here: the device-vs-host log10 sweep behind DESIGN.md's "known residuals", and two processes sharing one GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal
from oracle.binding import Oracle
from rnnoise_amd import capi, synth
from test_gpu_parity import oracle_run

pytestmark = pytest.mark.gpu


def _tiled_check(blob, N, calls, silent_stream):
    """N streams = N/32 replicas of a 32-stream block, fed from HBM through rnnoise_batch_process_device in `calls` (frames per
    call) on the batch's DEFAULT schedule -- what bench.py times: multi-frame calls run the three-stream frame pipeline
    (high-pass two frames ahead, analysis(f+1) beside network + synthesis(f)), the 6-slot pitch ring and the 3 spectra slots
    wrap, call boundaries fall inside the ring, the network runs layer-wise from 10,240 streams up and the analysis kernel
    four streams per workgroup.  A one-frame call in the middle takes the unpipelined route through the same state."""
    import torch
    T = sum(calls)
    base = synth.batch_pcm(range(32), T)
    base[:3, silent_stream] = 0                       # silent for 3 frames, then live
    base[T - 5:T - 3, (silent_stream + 3) % 32] = 0   # another one goes silent across the last call boundary region
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(base).to(dev).repeat(1, N // 32, 1).contiguous()
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty((T, N), device=dev)
    d_gains = torch.empty((T, N, 32), device=dev)
    m = capi.Model(blob)
    b = capi.Batch(m, N)
    assert b.set_nn_path(1) == 1                      # MFMA path is the default at this size
    st = torch.cuda.current_stream().cuda_stream
    f = 0
    for n in calls:
        b.process_device(d_out[f].data_ptr(), d_in[f].data_ptr(), d_vad[f].data_ptr(), d_gains[f].data_ptr(), n, st)
        f += n
    torch.cuda.synchronize()
    for name, t, w in (("pcm", d_out, 480), ("gains", d_gains, 32), ("vad", d_vad, 1)):
        r = t.view(torch.int32).reshape(T, N // 32, 32 * w)
        assert bool((r == r[:, :1]).all().item()), f"replicated streams diverged ({name})"
    out, gains, vad = d_out[:, :32].cpu().numpy(), d_gains[:, :32].cpu().numpy(), d_vad[:, :32].cpu().numpy()
    del d_in, d_out, d_gains, d_vad
    want = oracle_run(blob, base)
    assert want["silence"][:, silent_stream].any() and not want["silence"][:, 0].any()
    assert_bits_equal(out, want["out"], "pcm")
    assert_bits_equal(gains, want["gains"], "gains")
    assert_bits_equal(vad, want["vad"], "vad")
    for s_ in (0, 15, 16, silent_stream, 31):
        assert_bits_equal(b.export_state(N - 32 + s_), want["state"][s_], f"state of stream {N - 32 + s_}")
    # a stream from the middle of the batch as well (different XCD / tile than the first and the last block)
    mid = (N // 2 // 32) * 32
    assert_bits_equal(b.export_state(mid + 7), want["state"][7], f"state of stream {mid + 7}")
    b.close()
    m.close()


@pytest.mark.parametrize("rcp_profile", ["intel", "host"], indirect=True)
def test_65536_stream_batch_properties(blob_default, rcp_profile):
    """BASELINE configs[2] = the bench default: 14 frames as calls of 5 + 1 + 8 -- on the profile of the committed goldens and on
    "host", the profile a deployed process (and bench.py) runs on: the rcpps of this machine's CPU"""
    _tiled_check(blob_default, 65536, (5, 1, 8), silent_stream=21)


@pytest.mark.parametrize("rcp_profile", ["intel", "host"], indirect=True)
def test_sparser_model_32768(blob_little, rcp_profile):
    """BASELINE configs[3]: the sparser blob at 32,768 streams, 14 frames as calls of 5 + 1 + 8 -- on the goldens' profile and on
    the profile a deployed process runs on (this machine's rcpps)"""
    _tiled_check(blob_little, 32768, (5, 1, 8), silent_stream=9)


def _ragged_check(blob, N, calls, silent_stream):
    """_tiled_check for a batch size that is NOT a multiple of anything: N = 32 q + r streams = q replicas of a 32-stream block and
    the first r streams of one more.  Every kernel of the default schedule then has a partial unit at the end of its grid -- the
    lane = stream high-pass a partial wave, the four-stream analysis workgroups a partial workgroup, the front kernel a partial
    16-stream tile, the four-wave GRU layer kernel and the dense kernel a partial 64-stream group -- whose surplus lanes work on
    clamped addresses and must store nothing.  Whole replicas are compared among themselves, the ragged tail with the head of
    block 0, block 0 with the oracle; states are exported from the tail."""
    import torch
    T, q, r = sum(calls), N // 32, N % 32
    assert r and N % 64 and N % 16 and N % 4
    base = synth.batch_pcm(range(32), T)
    base[:3, silent_stream] = 0
    base[T - 5:T - 3, (silent_stream + 3) % 32] = 0
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(base).to(dev).repeat(1, q + 1, 1)[:, :N].contiguous()
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty((T, N), device=dev)
    d_gains = torch.empty((T, N, 32), device=dev)
    m = capi.Model(blob)
    b = capi.Batch(m, N)
    assert b.set_nn_path(1) == 1
    st = torch.cuda.current_stream().cuda_stream
    f = 0
    for n in calls:
        b.process_device(d_out[f].data_ptr(), d_in[f].data_ptr(), d_vad[f].data_ptr(), d_gains[f].data_ptr(), n, st)
        f += n
    torch.cuda.synchronize()
    for name, t, w in (("pcm", d_out, 480), ("gains", d_gains, 32), ("vad", d_vad, 1)):
        t = t.reshape(T, N, w).view(torch.int32)
        whole = t[:, :32 * q].reshape(T, q, 32 * w)
        assert bool((whole == whole[:, :1]).all().item()), f"replicated streams diverged ({name})"
        assert bool((t[:, 32 * q:] == t[:, :r]).all().item()), f"the ragged tail differs from the head of block 0 ({name})"
    out, gains, vad = d_out[:, :32].cpu().numpy(), d_gains[:, :32].cpu().numpy(), d_vad[:, :32].cpu().numpy()
    del d_in, d_out, d_gains, d_vad
    want = oracle_run(blob, base)
    assert want["silence"][:, silent_stream].any() and not want["silence"][:, 0].any()
    assert_bits_equal(out, want["out"], "pcm")
    assert_bits_equal(gains, want["gains"], "gains")
    assert_bits_equal(vad, want["vad"], "vad")
    for s_ in (N - 1, N - r, N - r - 1, (N // 2 // 32) * 32 + silent_stream):   # the last stream, the tail's first, the last whole block's last
        assert_bits_equal(b.export_state(s_), want["state"][s_ % 32], f"state of stream {s_}")
    b.close()
    m.close()


@pytest.mark.rcp("host")
@pytest.mark.parametrize("which", ["default", "little"])
def test_ragged_40037_streams(blob_default, blob_little, which):
    """VERDICT r5 Weak #1: the kernels that are the at-size defaults -- the four-wave rn_nn_gru_kernel (more 64-stream groups than CUs:
    > 16,384 streams) and the lane-per-stream high-pass inside pipelined calls -- on a batch with a partial tile (40,037 = 16 x
    2,502 + 5), a partial group (64 x 625 + 37), a partial analysis workgroup and a partial high-pass wave; default and sparser blob,
    default schedule, 14 frames as calls of 5 + 1 + 8.  Arithmetic under test: src/nnet.c:65-94, src/denoise.c:409-419."""
    _ragged_check(blob_default if which == "default" else blob_little, 40037, (5, 1, 8), silent_stream=3)


@pytest.mark.rcp("host")
def test_ragged_10277_streams_just_above_the_network_switch(blob_default):
    """The layer-wise network starts at 10,240 streams (batch.cpp: nn_layers_min_streams; 16,384 until round 6's last day): a
    ragged batch just above the switch -- 10,277 = 16 x 642 + 5 = 64 x 160 + 37, the eight-wave layer kernel (fewer 64-stream groups
    than CUs) with a partial tile and a partial group -- and, below, the largest batches of the tile kernel on both sides of 8,192
    streams (two tiles per CU, then three).  Arithmetic under test: src/nnet.c:65-94, src/rnn.c:44-60."""
    _ragged_check(blob_default, 10277, (4, 1, 3), silent_stream=3)


@pytest.mark.rcp("host")
@pytest.mark.parametrize("n", [8192, 10208])
def test_the_tile_network_at_its_largest_batches(blob_default, n):
    _tiled_check(blob_default, n, (3, 1, 2), silent_stream=9)


@pytest.mark.rcp("host")
def test_16384_streams_on_the_host_profile(blob_default):
    """the largest batch on the eight-wave layer kernel (one 64-stream group per CU), on the profile a deployed process gets by
    default (this CPU's rcpps)"""
    _tiled_check(blob_default, 16384, (3, 4), silent_stream=5)


def _soak(blob, N, reps, cycles):
    """The pipelined schedule for a long time: `reps` runs from reset of `cycles` x 24 frames each, as calls of 8 + 5 + 1 + 8 + 2
    frames over a 24-frame input that repeats (what fits in HBM at 65,536 streams), N/32 replicas of a 32-stream block.  After
    EVERY call every replica's pcm / gains / vad is compared with replica 0 on the GPU; replica 0's block is collected and, at the
    end of a run, compared frame by frame with one oracle run of the whole sequence, as is the state of streams from three
    places of the batch.  This is the test that would have seen round 4's "one run in ten" corruption in the product's own
    configuration (profiles/r5_gru_race.txt): a wrong bit anywhere in ~10^8 stream-frames fails it."""
    import torch
    calls, P = (8, 5, 1, 8, 2), 24
    T = cycles * P
    base = synth.batch_pcm(range(32), P)
    base[:3, 9] = 0                                   # a stream that is silent whenever the input wraps ...
    base[P - 5:P - 3, 12] = 0                         # ... and one that goes silent inside a call
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(base).to(dev).repeat(1, N // 32, 1).contiguous()
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty((P, N), device=dev)
    d_gains = torch.empty((P, N, 32), device=dev)
    m = capi.Model(blob)
    b = capi.Batch(m, N)
    assert b.set_nn_path(1) == 1
    st = torch.cuda.current_stream().cuda_stream
    want = oracle_run(blob, np.concatenate([base] * cycles))
    for rep in range(reps):
        b.reset()
        got = {"out": [], "gains": [], "vad": []}
        for c in range(cycles):
            f = 0
            for n in calls:
                b.process_device(d_out[f].data_ptr(), d_in[f].data_ptr(), d_vad[f].data_ptr(), d_gains[f].data_ptr(), n, st)
                for name, t, w in (("out", d_out, 480), ("gains", d_gains, 32), ("vad", d_vad, 1)):
                    r = t[f:f + n].view(torch.int32).reshape(n, N // 32, 32 * w)
                    assert bool((r == r[:, :1]).all().item()), f"run {rep}, frames {c * P + f}..{c * P + f + n - 1}: replicas diverged ({name})"
                    got[name].append(t[f:f + n, :32].cpu().numpy())
                f += n
        for name in got:
            assert_bits_equal(np.concatenate(got[name]), want[name], f"run {rep}: {name} of the first block, {T} frames")
        for s_ in (9, 31, (N // 2 // 32) * 32 + 12, N - 32 + 20):
            assert_bits_equal(b.export_state(s_), want["state"][s_ % 32], f"run {rep}: state of stream {s_} after {T} frames")
    b.close()
    m.close()


def test_soak_65536_streams(blob_default):
    """BASELINE configs[2]'s batch for 5 x 312 frames (10^8 stream-frames) on the default three-stream schedule"""
    _soak(blob_default, 65536, reps=5, cycles=13)


def test_soak_sparser_model_32768(blob_little):
    """the sparser blob at 32,768 streams (the configuration in which round 4 saw replicas diverge), 5 x 312 frames"""
    _soak(blob_little, 32768, reps=5, cycles=13)


@pytest.mark.parametrize("variant", [0, 1, 3])
def test_register_fft_bit_exact(variant):
    """F1 in isolation: the register-resident 960-point transform of the analysis / synthesis kernels (fft_reg.h; reference
    rnn_fft_c, src/kiss_fft.c:518-586) against the oracle's FFT on random, impulse, DC, alternating, tiny (denormal products)
    and huge inputs.  variant 0 = every exchange through ds_bpermute, 1 = DPP / swizzle forms, 3 = what the kernels use: as 1 with the
    lane ^ 32 level on v_permlane32_swap."""
    import ctypes as C
    rng = np.random.default_rng(960 + variant)
    cases = [(rng.standard_normal((960, 2)) * 3000).astype(np.float32) for _ in range(6)]
    imp = np.zeros((960, 2), np.float32); imp[0, 0] = 1; cases.append(imp)
    imp2 = np.zeros((960, 2), np.float32); imp2[517, 1] = -32768; cases.append(imp2)
    cases.append(np.full((960, 2), 12345.678, np.float32))                                   # DC
    alt = np.zeros((960, 2), np.float32); alt[::2, 0] = 1; alt[1::2, 0] = -1; cases.append(alt)  # Nyquist
    cases.append((rng.standard_normal((960, 2)) * 1e-38).astype(np.float32))                 # denormal inputs and products
    cases.append((rng.standard_normal((960, 2)) * 1e30).astype(np.float32))                  # no overflow at 1e30 / 960 * 960
    real = (rng.standard_normal((960, 2)) * 8000).astype(np.float32); real[:, 1] = 0; cases.append(real)  # what the kernels feed it
    cases.append(np.zeros((960, 2), np.float32))
    x = np.ascontiguousarray(np.stack(cases))
    y = np.empty_like(x)
    fp = C.POINTER(C.c_float)
    with capi.instrumented() as L:  # the probe kernels live in the instrumented library (include/rnnoise_amd_debug.h)
        rc = L.rnnoise_amd_debug_fft(0, variant, y.ctypes.data_as(fp), x.ctypes.data_as(fp), len(cases), 1, None, None)
    assert rc == 0
    for i, c in enumerate(cases):
        want = Oracle.fft(c.reshape(-1)).reshape(960, 2)
        assert_bits_equal(y[i], want, f"fft case {i} variant {variant}")


def test_log10_device_equals_host_libm_for_every_float():
    """(float)log10(1e-2 + (double)Ex), src/denoise.c:383 -- the one libm call on the path.  The kernels restate the host libm's
    algorithm (rnnoise_amd/csrc/log10_glibc.h); here the DEVICE code is swept against the host libm over every float Ex in
    [0, +Inf] (2,139,095,041 arguments: the device forms them from their bit patterns, the oracle side compares in C on all host
    threads).  Beside it, for the record, how often the device library's own log10 (round 4's implementation) rounds differently."""
    from concurrent.futures import ThreadPoolExecutor
    if not capi.log10_model().endswith("glibc-fma"):
        pytest.skip(f"log10 model {capi.log10_model()}: this host's libm is not the modelled one")
    CH, END = 1 << 26, 0x7f800001
    Oracle.lib()
    n_bad = {0: 0, 1: 0}
    first_bad = {}
    with capi.instrumented() as L, ThreadPoolExecutor(max(2, (os.cpu_count() or 4))) as pool:
        assert capi.log10_model().endswith("glibc-fma")
        for model in (0, 1):
            for lo in range(0, END, CH):
                n = min(CH, END - lo)
                got = np.empty(n, np.float32)
                assert L.rnnoise_amd_debug_log_energy_range(0, capi._fp(got), None, lo, n, model) == 0
                parts = [(lo + o, got[o:o + (1 << 22)]) for o in range(0, n, 1 << 22)]
                for (first, _), (bad, where) in zip(parts, pool.map(lambda p: Oracle.log_energy_range_diff(*p), parts)):
                    n_bad[model] += bad
                    if bad and model not in first_bad:
                        first_bad[model] = where
    print(f"log10, every float Ex in [0, Inf] ({END} arguments): the restated host algorithm differs from the host libm in {n_bad[0]}; "
          f"the device library's log10 in {n_bad[1]}" + (f" (first at Ex bits {first_bad[1]:#x})" if 1 in first_bad else ""))
    assert n_bad[0] == 0, f"{n_bad[0]} results differ, first at Ex bits {first_bad[0]:#x}"


_WORKER = r"""
import sys, lzma, zlib, numpy as np
sys.path.insert(0, {root!r})
from rnnoise_amd import capi, synth
blob = lzma.decompress(open({root!r} + "/tests/golden/default.blob.xz", "rb").read())
ids = list(range({first}, {first} + 48))
pcm = synth.batch_pcm(ids, 10, lead_silence=1)
m = capi.Model(blob); b = capi.Batch(m, 48, device=0); b.set_nn_path(1)
for rep in range(3):
    b.reset()
    out, vad, gains = b.process(pcm)
print("CRC", zlib.crc32(out.tobytes()), zlib.crc32(gains.tobytes()), zlib.crc32(vad.tobytes()), flush=True)
b.close(); m.close()
"""


def test_two_processes_share_one_gpu(blob_default):
    """replica independence: two processes, each with its own capi.Batch on device 0, run concurrently (what N ranks on N
    GPUs do, squeezed onto the one GPU this box has) and both produce the oracle's bits"""
    import zlib
    procs = [subprocess.Popen([sys.executable, "-c", _WORKER.format(root=ROOT, first=f)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for f in (0, 48)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    for first, (so, _) in zip((0, 48), outs):
        pcm = synth.batch_pcm(range(first, first + 48), 10, lead_silence=1)
        want = oracle_run(blob_default, pcm, collect_state=False)
        crc = [int(x) for x in so.split("CRC")[1].split()]
        assert crc == [zlib.crc32(want["out"].tobytes()), zlib.crc32(want["gains"].tobytes()),
                       zlib.crc32(want["vad"].tobytes())], f"process starting at stream {first}"


def test_bench_with_two_ranks_on_one_device():
    """bench.py's own multi-rank path on real kernels where only one GPU exists: two ranks of torch.distributed.run, both on
    device 0 (RNNOISE_AMD_BENCH_SHARE_DEVICE=1: gloo carries the reduction, RCCL refuses two ranks on one device) -- disjoint
    stream shards, whole-job aggregate over both ranks, rank-0-only line, parity leg green.  Not a scaling number."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=ROOT, RNNOISE_AMD_BENCH_SHARE_DEVICE="1", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "4096", "--steps", "4",
                        "--warmup", "2", "--repeats", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["shared_device"] is True and d["scaling"] == "weak"
    assert d["config"]["streams_per_gpu"] == 4096 and d["config"]["frames_per_step"] == 8192
    assert d["config"]["stream_ids_by_rank"] == [[0, 4096], [4096, 8192]]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / 8192 - 1) < 0.01
    assert d["parity"]["bit_identical"] is True and d["config"]["outputs_sane"] is True
    assert "cpu_baseline" not in d


def test_bench_falls_back_to_gloo_when_rccl_refuses():
    """The real RCCL in the failure path of bench.py's collectives (open_collectives): two ranks whose launcher shows each of them ONE
    device -- the same one (HIP_VISIBLE_DEVICES=0) -- both ask RCCL for a communicator, RCCL refuses ("Duplicate GPU detected"), every
    rank votes over the gloo control group, and the job runs to its line on gloo: "collective": "gloo-fallback" with the first error
    text, the aggregate over both ranks, parity green.  What the first multi-GPU run does if its communicator cannot be built."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=ROOT, HIP_VISIBLE_DEVICES="0", RNNOISE_AMD_BENCH_DATA_TIMEOUT="30", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RNNOISE_AMD_BENCH_SHARE_DEVICE", "RNNOISE_AMD_BENCH_DATA_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "4096", "--steps", "4",
                        "--warmup", "2", "--repeats", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    if d["collective"] == "rccl":
        pytest.skip("this RCCL accepts two ranks on one device: nothing to fall back from")
    assert d["collective"] == "gloo-fallback" and d["collective_error"], d
    assert d["n_gpus"] == 2 and len(d["value_by_rank"]) == 2
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / 8192 - 1) < 0.01
    assert d["parity"]["bit_identical"] is True and d["config"]["outputs_sane"] is True
