#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X rnnoise_process_frame() path.

Metric (BASELINE.json): 10 ms frames/s (48 kHz mono) summed over N concurrent streams, and
the fraction of the HBM roofline.  A "step" is one pass of the hot path over one batch: every
stream of the batch advances by one 480-sample frame (high-pass -> analysis -> network ->
synthesis).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--model default|little]
                  [--nn mfma|vector] [--repeats R] [--host-io] [--s16] [--frames-per-call F]

Workload at N=1: BASELINE.json configs[2] -- 65,536 concurrent streams on one MI355X (the largest
single-GPU configuration), default architecture, int8 model, network recast as batched MFMA
GEMMs.  `--streams 4096 [--nn vector]` is configs[1], `--model little --streams 32768` is
configs[3].  With --gpus N every rank owns its own S streams (N x 65,536 = configs[4] at N=8:
independent streams shard trivially, SURVEY 8e: "weak" scaling, no data-path collective); the only
collectives are the barriers and the max-over-ranks of the elapsed times.

`python bench.py --gpus N` may be started directly (it re-executes itself under
torch.distributed.run, one rank per GPU) or under a launcher that already set RANK / WORLD_SIZE.

Timing: W untimed warm-up steps, then R repetitions (default 25) of EXACTLY K steps, each
bracketed by barrier + synchronize on both sides; every repetition's time is the MAX over
ranks; `value` is the median repetition (min / max are reported beside it) and
`ms_per_step x steps` is that repetition.

The model is the synthetic default-architecture blob produced by the reference's own
exporter (the trained weights are a separate download upstream; tests/golden/make_golden.py).
Input PCM is synthetic (rnnoise_amd/synth.py recipe, evaluated on the GPU with torch) and is
resident in HBM before the timed region starts; no silent frames, so the network runs on
every frame of every stream.
"""
from __future__ import annotations

import argparse
import json
import lzma
import math
import os
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, MI355X spec (MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9    # max shader clock (same guide)
N_SIMD = 256 * 4    # CUs x SIMDs
FRAME = 480
METRIC = "10ms frames/sec (48kHz mono) at N concurrent streams; % HBM roofline"

# Algorithmic HBM bytes per stream-frame of each kernel; W is added to the network
# kernel per LAUNCH from the model: the weights are read from HBM at most once per launch.
# (round 6: pitch_buf is kept 2x decimated beside the ring -- rn_dev.h RN_XRING_SLOT -- so K0's autocorrelation pass reads the 624 OLD
#  decimated samples where it read 1728 ring samples, the frame's own 240 going from the biquad's registers straight into the chains, and
#  K1's downsampling reads 864 floats where it read 1728: 3,444 fewer bytes for K0, 3,448 fewer for K1 than in rounds 1-5)
#   K0: input 1920 r + ring slot 1920 w + decimated slot 960 w + hp state 8 r + 8 w + last sample of the previous slot 4 r
#       + old decimated samples 2496 r + pitch_buf[0..1] 8 r (autocorrelation) + 5 taps 20 w
#   K1: decimated pitch_buf 3456 + pitch_buf[0..1] 8 (downsample) + 2 x 3840 (windows) r, X re-read 3200 r, taps 20 r;
#       X,P 7696 + E 384 + features 260 + flags 12 w
#   K2: features 260 r, conv/GRU state 2 x (520 + 1024 + 4608), gains 128 + vad 4 w
#   K3: X,P 7696 + E 2 x 384 + gains 128 + lastg 2 x 128 + synth_mem 2 x 1920 r/w + out 1920 w
ALG_BYTES = {
    "highpass": 1920 + 1920 + 960 + 16 + 4 + 2496 + 8 + 20,
    "analysis": (3456 + 8 + 3840 + 3840 + 3200 + 20) + (7696 + 384 + 260 + 12),
    "network": 260 + 2 * (520 + 1024 + 4608) + 128 + 4,
    "synthesis": 3848 + 3848 + 384 + 128 + 128 + 256 + 1920 + 1920 + 1920,
}
KERNEL_OF = {"highpass": "rn_hp_kernel", "analysis": "rn_analysis_kernel", "network": "rn_nn_mfma_kernel",
             "synthesis": "rn_synthesis_kernel"}
# batch.cpp nn_layers_min_streams() / nn_one_max_streams(), hp_kernel.hip RN_HP_ONE_MAX, dsp_kernels.hip RN_K1_MULTI_MIN_STREAMS:
# the batch sizes at which the library switches kernels (it honours the same environment variables)
NN_LAYERS_MIN_STREAMS = int(os.environ.get("RNNOISE_AMD_NN_LAYERS_MIN", "10240"))
NN_ONE_MAX_STREAMS = int(os.environ.get("RNNOISE_AMD_NN_ONE_MAX", "512"))
HP_ONE_MAX_STREAMS = int(os.environ.get("RNNOISE_AMD_HP_ONE_MAX", "2048"))  # (hp_kernel.hip: RN_HP_ONE_MAX, pipelined and one-frame calls)
K1_SPW_FORCE = int(os.environ.get("RNNOISE_AMD_K1_SPW", "0"))
TILE_WAVES_FORCE = int(os.environ.get("RNNOISE_AMD_TILE_WAVES", "0"))
K1_MULTI_MIN_STREAMS = 2560
K3_FEW_MAX_STREAMS = 256
N_CU = 256
NN_LAYER_KERNELS = ("rn_nn_front_kernel", "rn_nn_gru_kernel", "rn_nn_gru_kernel", "rn_nn_gru_kernel", "rn_nn_dense_kernel")


def kernel_of(kind: str, n_streams: int, nn: str = "mfma", alone: bool = False) -> str:
    """Name of the kernel behind a timed kind (rocprofv3 / PMC tables use it).  The network kind of a large batch is five
    launches timed as one (front, three GRU layers, dense); its PMC record is that of the GRU layer kernel, which is
    three of the five and the longest."""
    if kind == "network":
        if nn != "mfma":
            return "rn_nn_one_kernel" if n_streams <= NN_ONE_MAX_STREAMS else "rn_nn_vector_kernel"
        if n_streams < NN_LAYERS_MIN_STREAMS:
            # (nn_mfma.hip: sixteen waves per tile in one-frame calls while every tile has a CU to itself; no PMC pass records that form)
            sixteen = TILE_WAVES_FORCE == 16 or (TILE_WAVES_FORCE != 8 and alone and -(-n_streams // 16) <= N_CU)
            return "rn_nn_mfma16_kernel" if sixteen else "rn_nn_mfma_kernel"
        # (nn_layers.hip: the four-wave form once there are more 64-stream groups than CUs, the eight-wave one below)
        return "rn_nn_gru_kernel" if -(-n_streams // 64) > N_CU else "rn_nn_gru_w8_kernel"
    if kind == "analysis" and k1_single(n_streams):
        return "rn_analysis_single_kernel"  # one stream per workgroup
    if kind == "highpass" and n_streams <= HP_ONE_MAX_STREAMS:
        return "rn_hp_one_kernel"           # one wave per stream
    if kind == "synthesis" and n_streams <= K3_FEW_MAX_STREAMS:
        return "rn_synthesis_few_kernel"    # every operand requested up front (dsp_kernels.hip: RN_K3_FEW_MAX)
    return KERNEL_OF[kind]


def k1_single(n_streams: int) -> bool:
    return K1_SPW_FORCE == 1 or (K1_SPW_FORCE == 0 and n_streams < K1_MULTI_MIN_STREAMS)


def waves_per_launch(kind: str, n_streams: int, nn: str = "mfma") -> int:
    if kind == "network" and nn != "mfma":
        return n_streams * (14 if n_streams <= NN_ONE_MAX_STREAMS else 6)  # rn_nn_one_kernel: 14 waves per stream; vector: 384 threads
    return {"highpass": n_streams if n_streams <= HP_ONE_MAX_STREAMS else -(-n_streams // 64),
            "analysis": n_streams if k1_single(n_streams) else -(-n_streams // 4) * 4,
            "network": (-(-n_streams // 64) if n_streams >= NN_LAYERS_MIN_STREAMS else -(-n_streams // 16)) * 8,
            "synthesis": n_streams}[kind]


def valu_cost(kernel: str) -> float:
    """Mean clocks per wave64 VALU instruction per SIMD of `kernel`: its instruction mix (tools/valu_mix.py over the built
    objects -> profiles/valu_mix.json) priced with the per-instruction issue costs MEASURED on the MI355X
    (profiles/r4_valu_issue.txt: 2.26 clk for plain f32/u32 VOP2 forms, 4.15 for DPP / SGPR-source / packed / f64 / compare /
    select / convert / 3-operand forms, 8.12 for transcendentals and v_permlane32_swap).  MI355X_MICROARCH.md's "2 cycles per
    wave64 v_fma_f32" holds for the first class only; the PMC ratio 4*SQ_ACTIVE_INST_VALU/SQ_INSTS_VALU (4.0 for every
    kernel) is a property of the counter.  4.15 when the kernel has no record."""
    try:
        with open(os.path.join(ROOT, "profiles", "valu_mix.json")) as f:
            return float(json.load(f)["kernels"][kernel]["mean_cycles"])
    except Exception:
        return 4.15


def valu_cost_kind(kernel: str) -> str:
    """"dynamic" when the kernel's mean cost weights its code sections by MEASURED per-section instruction counts (tools/valu_mix.py
    --k1-sections: the analysis kernel, whose loops execute a mix that is not the file's average), "static" otherwise."""
    try:
        with open(os.path.join(ROOT, "profiles", "valu_mix.json")) as f:
            return "dynamic" if str(json.load(f)["kernels"][kernel].get("weighting", "static")).startswith("dynamic") else "static"
    except Exception:
        return "static"


def clock_hz(rec) -> float:
    """Shader clock a kernel RAN at: GRBM_GUI_ACTIVE / duration of its PMC pass (profiles/pmc_by_streams.json: clock_ghz);
    the guide's 2.4 GHz maximum when the record carries no measurement."""
    return float(rec["clock_ghz"]) * 1e9 if rec and rec.get("clock_ghz") else CLOCK_HZ


VALU_FLOOR = 2.26  # clocks per wave64 instruction if every one were a plain f32 VOP2 (the guide's vector-f32 peak, as measured)


def step_valu_issue_ms(n_streams: int, model: str = "default", nn: str = "mfma"):
    """VALU issue time of one frame step if every SIMD issued back to back: over the step's kernels, waves x VALU instructions
    per wave (PMC, profiles/pmc_by_streams.json) x valu_cost(kernel), spread over 1024 SIMDs at 2.4 GHz.
    None when a kernel of the step has no PMC record (the vector network path)."""
    n = n_streams
    launches = [(kernel_of("highpass", n, nn), waves_per_launch("highpass", n)), (kernel_of("analysis", n, nn), waves_per_launch("analysis", n)),
                ("rn_synthesis_kernel", n)]
    if nn != "mfma":
        return None
    if n >= NN_LAYERS_MIN_STREAMS:
        launches += [("rn_nn_front_kernel", -(-n // 16) * 8), ("rn_nn_gru_kernel", 3 * (-(-n // 64)) * 4), ("rn_nn_dense_kernel", -(-n // 64) * 8)]
    else:
        launches += [("rn_nn_mfma_kernel", -(-n // 16) * 8)]
    cycles = 0.0  # (seconds x SIMDs: every kernel at the clock it was measured at)
    for kernel, waves in launches:
        r = pmc_record(kernel, n, model)
        if not r or "valu_per_wave" not in r:
            return None
        cycles += waves * r["valu_per_wave"] * valu_cost(r.get("kernel", kernel)) / clock_hz(r)
    return 1e3 * cycles / N_SIMD


def load_blob(name: str = "default") -> bytes:
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.blob.xz"), "rb") as f:
        return lzma.decompress(f.read())


def synth_pcm_torch(torch, n_streams: int, n_frames: int, device, seed_base: int):
    """(T, N, 480) float32 on `device`: the SURVEY 8d signal (harmonics + filtered noise), s16-rounded."""
    n = n_frames * FRAME
    t = torch.arange(n, device=device, dtype=torch.float32) / 48000.0
    sid = torch.arange(n_streams, device=device, dtype=torch.float32)[:, None] + seed_base
    out = torch.empty((n_frames, n_streams, FRAME), device=device, dtype=torch.float32)
    g = torch.Generator(device=device)
    g.manual_seed(20250223 + seed_base)
    chunk = max(1, min(n_streams, (64 << 20) // n))
    for s0 in range(0, n_streams, chunk):
        ids = sid[s0:s0 + chunk]
        f0 = (90.0 + torch.remainder(ids, 160.0)) + 40.0 * torch.sin(2 * torch.pi * 0.5 * t)[None, :]
        phi = 2 * torch.pi * torch.cumsum(f0.double(), dim=1).float() / 48000.0
        h = torch.zeros_like(phi)
        for k in range(1, 20):
            h += torch.sin(k * phi) / k
        h *= ((0.5 + 0.5 * torch.sin(2 * torch.pi * 1.3 * t)) ** 2)[None, :]
        h /= h.abs().amax(dim=1, keepdim=True).clamp_min(1e-9)
        w = torch.randn((ids.shape[0], n + 7), device=device, generator=g)
        nz = torch.nn.functional.avg_pool1d(w[:, None, :], 8, 1)[:, 0, :]
        x = torch.clamp(torch.round(6000.0 * h + 1500.0 * nz), -32768, 32767)
        out[:, s0:s0 + chunk] = x.reshape(ids.shape[0], n_frames, FRAME).permute(1, 0, 2)
    return out


def pmc_record(kernel: str, n_streams: int, model: str = "default"):
    """Per-kernel counters from the committed PMC passes (profiles/pmc_by_streams.json, written by
    tools/make_profile_tables.py from rocprofv3 --pmc runs: FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU in separate passes,
    gfx950 x2 read correction): the set measured nearest (in octaves) to this batch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_by_streams.json")) as f:
            allsets = json.load(f)
        sets = allsets.get(model) if model != "default" and allsets.get(model) else allsets["by_streams"]
        # (the nearest set that HAS the kernel: which form of a kernel a batch size runs also depends on the schedule of the pass)
        for n in sorted(sets, key=lambda n: (abs(math.log2(int(n) / n_streams)), -int(n))):
            k = sets[n]
            r = k.get(kernel) or k.get(kernel.replace("_lean", "").replace("_single", "").replace("_w8", ""))
            if r:
                return r
        return None
    except Exception:
        return None


def cgroup_cpu_quota():
    """CPUs the cgroup allows (cpu.max / cfs quota), or None when unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(math.ceil(int(q) / int(p))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, int(math.ceil(q / p)))
    except Exception:
        pass
    return None


def usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cgroup_cpu_quota()
    return max(1, min(n, q) if q else n)


def cpu_baseline(blob: bytes):
    """The reference itself (oracle/_ref, kind "reference") or our restatement (kind "port") on the host cores:
    a 1-thread leg (~4 s) and an all-usable-cores leg (~10 s); see oracle/cpu_bench.c."""
    import numpy as np
    from rnnoise_amd import synth
    ref = os.path.join(ROOT, "oracle", "_ref", "cpu_bench_ref")
    port = os.path.join(ROOT, "oracle", "cpu_bench_port")
    cores = usable_cpus()  # (both timers are built by __graft_entry__.build(): nothing is compiled inside the benchmark)
    with tempfile.TemporaryDirectory() as td:
        bp, pp = os.path.join(td, "m.blob"), os.path.join(td, "pcm.s16")
        open(bp, "wb").write(blob)
        np.concatenate([synth.stream_pcm(s, 200) for s in range(8)]).tofile(pp)
        for exe, kind in ((ref, "reference"), (port, "port")):
            if not os.path.exists(exe):
                continue
            try:
                legs = []
                for threads, secs in ((1, 4), (cores, 10)):
                    r = subprocess.run([exe, bp, pp, str(threads), str(secs)], capture_output=True, text=True, timeout=120)
                    legs.append(json.loads(r.stdout.strip().splitlines()[-1]))
            except Exception:
                continue
            cpu = "unknown"
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        cpu = line.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            one, many = legs
            return {"value": round(many["frames_per_s"], 1), "unit": "frames/s", "cores": many["threads"], "kind": kind,
                    "frames_per_s_per_core": round(many["frames_per_s"] / many["threads"], 1),
                    "one_thread_frames_per_s": round(one["frames_per_s"], 1),
                    "sample": f"{many['frames']} frames in {many['seconds']:.1f} s on {many['threads']} threads (one stream "
                              f"each; threads = min(sched_getaffinity {len(os.sched_getaffinity(0))}, cgroup quota "
                              f"{cgroup_cpu_quota()}), os.cpu_count {os.cpu_count()}) + {one['frames']} frames in "
                              f"{one['seconds']:.1f} s on 1 thread; 200-frame synthetic PCM looped in memory, same blob; "
                              f"host CPU {cpu}"}
    return None


def parity_leg(capi, torch, batch, blob, d_in, n_frames: int, s16: bool):
    """Checker leg (SURVEY 8d "parity summary of the same run"), outside the timed region: the batch that was just timed
    is reset and fed the first `n_frames` frames of the bench's own input in ONE call (default schedule, the pipelined
    route the timed region took); 16 streams from the start and 16 from the end of the batch are compared bit for bit --
    PCM, VAD, raw gains -- with the oracle (oracle/, the CPU restatement pinned to the compiled reference) run on the same
    samples under the rcpps profile the library is using."""
    import numpy as np
    from oracle import binding
    N = batch.n
    prof = capi.rcp_profile()
    binding.set_rcp_profile("host" if prof.startswith("host") else prof)
    sample = list(range(min(16, N))) + list(range(max(16, N - 16), N) if N > 16 else [])
    x = d_in[:n_frames]
    dev = x.device
    out = torch.empty_like(x, dtype=torch.int16 if s16 else torch.float32)
    xin = x.to(torch.int16) if s16 else x.contiguous()
    vad = torch.empty((n_frames, N), device=dev)
    gains = torch.empty((n_frames, N, 32), device=dev)
    dbg = (lambda m: print(f"[parity] {m}", file=sys.stderr, flush=True)) if os.environ.get("BENCH_TRACE") else (lambda m: None)
    dbg("allocated")
    batch.reset()
    dbg("reset")
    batch.process_device(out.data_ptr(), xin.data_ptr(), vad.data_ptr(), gains.data_ptr(), n_frames,
                         torch.cuda.current_stream().cuda_stream, s16=s16)
    torch.cuda.synchronize()
    dbg("processed")
    idx = torch.tensor(sample, device=dev)
    got_o, got_v, got_g = out[:, idx].cpu().numpy(), vad[:, idx].cpu().numpy(), gains[:, idx].cpu().numpy()
    pcm = x[:, idx].cpu().numpy()
    bad = 0
    for j in range(len(sample)):
        want = binding.Oracle(blob).run(pcm[:, j])
        wo = want["out"]
        if s16:  # examples/rnnoise_demo.c:58, x86 conversion: cvttss2si, low 16 bits
            wi = np.where((wo >= -2147483648.0) & (wo < 2147483648.0), np.trunc(wo), -2147483648.0).astype(np.int64)
            same_o = np.array_equal(got_o[:, j], (wi & 0xFFFF).astype(np.uint16).view(np.int16))
        else:
            same_o = np.array_equal(got_o[:, j].view(np.uint32), wo.view(np.uint32))
        if not (same_o and np.array_equal(got_v[:, j].view(np.uint32), want["vad"].view(np.uint32))
                and np.array_equal(got_g[:, j].view(np.uint32), want["gains"].view(np.uint32))):
            bad += 1
    return {"streams": len(sample), "frames": n_frames, "bit_identical": bad == 0, "streams_differing": bad,
            "batch_streams": N, "checked": "pcm" + (" (s16)" if s16 else "") + ", vad, raw gains", "rcp_profile": prof,
            "against": "oracle/rn_oracle.c (pinned to the compiled reference)"}


class StubBatch:
    """Launcher self-test only (--stub, CPU + gloo; tests/test_bench_launcher_cpu.py): stands in for capi.Batch so that
    the rank / shard / aggregate / rank-0-JSON logic of THIS file can run where no GPU exists.  It computes nothing; the
    line it yields is marked "stub": true and carries a metric name no report can mistake for a measurement."""

    def __init__(self, n):
        self.n = n

    def set_nn_path(self, p):
        return 1

    def process_device(self, *a):
        time.sleep(2e-4)

    def enable_timing(self, on=True):
        pass

    def kernel_ms(self):
        return dict(analysis=0.3, network=0.2, synthesis=0.1, highpass=0.05, launches=1)


def pin_rank_cpus(local_rank: int, local_world: int, torch) -> None:
    """One node, several ranks: each rank keeps to its own slice of the CPUs this job may use and sizes its thread pools
    for its share of the cgroup quota, so that eight ranks importing torch and launching kernels at once do not fight over
    (say) sixteen CPUs of quota with sixteen threads each.  A no-op for one rank."""
    if local_world <= 1:
        return
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // local_world
        share = usable_cpus() // local_world  # (this rank's part of the job's quota: taken BEFORE the affinity is narrowed)
        if per >= 1:
            os.sched_setaffinity(0, cpus[local_rank * per:(local_rank + 1) * per])
        torch.set_num_threads(max(1, min(per, share)))
    except (AttributeError, OSError):
        pass


def device_report(torch, rank: int, local_rank: int, gpu_index: int, stub: bool) -> dict:
    """What this rank can say about the device it runs on: torch's index and name, the PCI address (the one identity a device mask
    cannot rename), the library's own device count, and the masks in the environment."""
    rep = {"rank": rank, "local_rank": local_rank, "device": None if stub else gpu_index,
           "masks": {v: os.environ[v] for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if v in os.environ}}
    if stub:
        return rep
    try:
        p = torch.cuda.get_device_properties(gpu_index)
        rep["name"] = p.name
        if hasattr(p, "pci_bus_id"):
            rep["pci_bus"] = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}"
        rep["torch_device_count"] = torch.cuda.device_count()
        from rnnoise_amd import capi
        rep["rnnoise_amd_device_count"] = int(capi.lib().rnnoise_amd_device_count())
    except Exception as e:  # noqa: BLE001 -- a report, not a requirement
        rep["error"] = repr(e)[:200]
    return rep


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(a) -> int:
    """`python bench.py --gpus N` started directly: become N ranks of torch.distributed.run on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    return subprocess.call(cmd, env=env)


def workload_name(a, n_streams: int) -> str:
    cfg = {("default", 65536): "configs[2]", ("default", 4096): "configs[1]", ("little", 32768): "configs[3]"}.get(
        (a.model, n_streams), "custom size")
    if a.gpus == 8 and a.model == "default" and n_streams == 65536:
        cfg = "configs[4] (8 x configs[2])"
    dens = "density 1/3" if a.model == "default" else "sparser blob (rnnoise_data_little stand-in)"
    return (f"BASELINE {cfg}: {n_streams} concurrent streams per GPU, default architecture (conv 65x3->128->384, "
            f"3xGRU(384) block-sparse int8, {dens}), synthetic exporter-made model, network path = {a.nn}"
            + (", host-fed (pinned, double-buffered PCIe)" if a.host_io else "") + (", int16 PCM at both ends" if a.s16 else "")
            + (f", {a.frames_per_call} frame(s) per call" if a.frames_per_call else ""))


def open_collectives(torch, rank: int, world: int, dev, data_backend):
    """The job's two process groups.  CONTROL = gloo, always (rendezvous, object gathers, the agreement below): it needs nothing
    but TCP on 127.0.0.1.  DATA = RCCL over xGMI for the barriers around the timed regions and the MAX all-reduce of the times
    (north_star: "RCCL only for the final throughput reduction") -- created and TRIED here, once, with a timeout, and every rank
    votes over gloo on whether its try worked: one rank without a working communicator (or all of them) sends the whole job to
    gloo for the data group as well, and the line says so ("collective": "gloo-fallback" + the first error) instead of the run
    dying, or hanging, in its first collective.  Returns (dist, data_group, "rccl" | "gloo" | "gloo-fallback", error text or None)."""
    import datetime

    import torch.distributed as dist
    # a failed or timed-out RCCL collective must RAISE here, not abort the process from the watchdog thread
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    if not data_backend:
        return dist, None, "gloo", None
    group, err = None, None
    try:
        if os.environ.get("RNNOISE_AMD_BENCH_BREAK_RCCL") == str(rank):  # (tests: this rank's communicator "fails")
            raise RuntimeError("RCCL failure injected by RNNOISE_AMD_BENCH_BREAK_RCCL")
        group = dist.new_group(backend=data_backend, timeout=datetime.timedelta(seconds=float(os.environ.get("RNNOISE_AMD_BENCH_DATA_TIMEOUT", "90"))))
        probe = torch.ones(1, device=dev if data_backend == "nccl" else "cpu")
        dist.all_reduce(probe, group=group)
        if data_backend == "nccl":
            torch.cuda.synchronize()
        if int(probe.item()) != world:
            raise RuntimeError(f"RCCL all-reduce of ones over {world} ranks returned {probe.item()}")
    except Exception as e:  # noqa: BLE001 -- whatever the runtime throws: the vote decides
        err = f"rank {rank}: {type(e).__name__}: {str(e)[:300]}"
        print(f"[bench] {err}", file=sys.stderr, flush=True)
    votes = [None] * world
    dist.all_gather_object(votes, err)
    bad = [v for v in votes if v]
    if bad:
        return dist, None, "gloo-fallback", bad[0]
    return dist, group, "rccl" if data_backend == "nccl" else data_backend, None


def bench_rank(a) -> dict | None:
    """One rank of the benchmark (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  Returns the JSON line's dict
    on rank 0, None elsewhere."""
    import torch
    from rnnoise_amd.dist import aggregate_times, shard_streams

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    a.gpus = world
    stub = a.stub
    if not stub and not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU path)")
    # RNNOISE_AMD_BENCH_SHARE_DEVICE=1 (tests on a one-GPU box): every rank runs its shard on device 0 and the reduction goes
    # over gloo, since RCCL refuses two ranks on one device; the line says so ("shared_device": true) and is not a scaling number
    share = os.environ.get("RNNOISE_AMD_BENCH_SHARE_DEVICE") == "1" and not stub
    gpu_index = 0 if share else local_rank
    if not stub and not share and local_rank > 0 and torch.cuda.device_count() == 1 and any(
            os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        gpu_index = 0  # (a launcher that masks the devices per rank: each rank sees its own GPU as device 0)
    dev = torch.device("cpu") if stub else torch.device("cuda", gpu_index)
    if not stub:
        if gpu_index >= torch.cuda.device_count():
            sys.exit(f"bench.py: rank {rank} (local rank {local_rank}) needs GPU {gpu_index}, but this node shows {torch.cuda.device_count()} "
                     f"device(s): start one rank per visible GPU (or RNNOISE_AMD_BENCH_SHARE_DEVICE=1 to test the rank path on one)")
        torch.cuda.set_device(gpu_index)
    pin_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), torch)
    dist, data_group, collective, collective_error = None, None, None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (tests: RNNOISE_AMD_BENCH_DATA_BACKEND=gloo sends the stub through the same create / try / vote code with a gloo data group)
        backend = os.environ.get("RNNOISE_AMD_BENCH_DATA_BACKEND") or (None if (stub or share) else "nccl")
        dist, data_group, collective, collective_error = open_collectives(torch, rank, world, dev, backend)
    red_dev = dev if collective == "rccl" else torch.device("cpu")  # where the reduction's tensors live

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist:
            if collective == "rccl":
                dist.barrier(group=data_group, device_ids=[gpu_index])  # (explicit: no guessing of the rank's device)
            else:
                dist.barrier(group=data_group)
        sync()

    # weak scaling: `streams` per GPU; this rank's global stream ids seed its signals
    N, K, Wm, R = a.streams, a.steps, a.warmup, a.repeats
    mine = shard_streams(N * world, world, rank)
    assert len(mine) == N
    blob = load_blob(a.model)
    if stub:
        W, batch, model = 1521668, StubBatch(N), None
        cap, d_in = 8, None
    else:
        from rnnoise_amd import capi
        model = capi.Model(blob)
        W = model.weight_bytes
        batch = capi.Batch(model, N, device=gpu_index)
        batch.set_nn_path(1 if a.nn == "mfma" else 0)
        # inputs resident in HBM; cycle through at most `cap` distinct frames if K+W is large
        cap = max(8, min(K + Wm, (3 << 30) // (N * FRAME * 4)))
        if a.host_io:
            cap = max(4, min(cap, (2 << 30) // (N * FRAME * (2 if a.s16 else 4))))  # pinned host memory: 2 GB each way at most
        d_in = d_in_f32 = synth_pcm_torch(torch, N, cap, dev, seed_base=mine.start)
        if a.s16:
            d_in = d_in.to(torch.int16)  # (the synthetic samples are s16-rounded already: exact)
        d_out = torch.empty_like(d_in)
        d_vad = torch.empty((cap, N), device=dev)
        d_gains = torch.empty((cap, N, 32), device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        if a.host_io:  # PCIe-inclusive: pinned host buffers, DMA in place, two chunks in flight inside the library
            h_in = d_in.cpu().pin_memory()
            h_out = torch.empty_like(h_in).pin_memory()
            h_vad = torch.empty((cap, N)).pin_memory()
    esz = N * FRAME * (2 if a.s16 else 4)

    def run(first: int, count: int):
        f, left = first, count
        while left > 0:
            k = f % cap
            n = min(left, cap - k, a.frames_per_call or left)
            if stub:
                batch.process_device()
            elif a.host_io:
                batch.process_into(h_out.data_ptr() + k * esz, h_in.data_ptr() + k * esz, h_vad.data_ptr() + k * N * 4, 0, n, s16=a.s16)
            else:
                batch.process_device(d_out.data_ptr() + k * esz, d_in.data_ptr() + k * esz, d_vad.data_ptr() + k * N * 4,
                                     d_gains.data_ptr() + k * N * 128, n, stream, s16=a.s16)
            f += n
            left -= n

    run(0, Wm)
    barrier()
    batch.enable_timing(True)
    times = []
    for r in range(R):
        barrier()
        t0 = time.perf_counter()
        run(Wm + r * K, K)
        barrier()
        times.append(time.perf_counter() - t0)
    if os.environ.get("BENCH_TRACE"):
        print("[bench] timed region done", file=sys.stderr, flush=True)
    kms = batch.kernel_ms()
    if os.environ.get("BENCH_TRACE"):
        print("[bench] kernel_ms read", file=sys.stderr, flush=True)
    # the same K steps once more with every kernel on ONE stream: stand-alone kernel durations (in the pipelined schedule
    # the kernels of neighbouring frames share the machine, which stretches each one's own duration)
    kms_alone = None
    if not stub and not a.host_io:
        old = batch.set_schedule(9)
        run(Wm + R * K, K)
        sync()
        batch.kernel_ms()
        run(Wm + (R + 1) * K, K)
        sync()
        kms_alone = batch.kernel_ms()
        batch.set_schedule(old)
    batch.enable_timing(False)
    own_median = statistics.median(times)
    times = aggregate_times(times, dist, red_dev, group=data_group)  # element-wise MAX over ranks: the one data-path collective
    shards, rank_medians = [[mine.start, mine.stop]], [own_median]
    if dist:  # (object gathers ride on the gloo control group)
        shards, rank_medians = [None] * world, [None] * world
        dist.all_gather_object(shards, [mine.start, mine.stop])
        dist.all_gather_object(rank_medians, own_median)
    # which device every rank really ran on: a launcher that masks or mis-numbers the GPUs shows up in the line itself (two ranks with
    # one PCI bus id = two ranks on one GPU, whatever their indices say)
    report = device_report(torch, rank, local_rank, gpu_index, stub)
    devices = [report]
    if dist:
        devices = [None] * world
        dist.all_gather_object(devices, report)
    frames_per_rep = float(N * K * world)
    med = statistics.median(times)
    line = None
    if rank == 0:
        sane = True
        if not stub:
            o, v = (h_out, h_vad) if a.host_io else (d_out, d_vad)
            sane = bool(torch.isfinite(o).all().item()) and float(v.max().item()) > 0.0
        kinds = ("highpass", "analysis", "network", "synthesis")
        per_launch = {k: ALG_BYTES[k] * N + (W if k == "network" else 0) for k in kinds}
        # dominant KERNEL = longest single launch: the layer-wise network is five launches whose durations kernel_ms adds up
        n_launch = {k: (len(NN_LAYER_KERNELS) if k == "network" and a.nn == "mfma" and N >= NN_LAYERS_MIN_STREAMS else 1) for k in kinds}
        dom = max(kinds, key=lambda k: kms[k] / n_launch[k])
        kname = kernel_of(dom, N, a.nn, alone=a.frames_per_call == 1)
        # a kind of several launches (the layer-wise network) is represented by its MEAN launch: a fifth of the kind's bytes in a
        # fifth of the kind's time (the library times the five launches separately but reports their sum)
        dom_ms = kms[dom] / n_launch[dom]
        dom_bytes = per_launch[dom] // n_launch[dom]
        ach = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        pmc = pmc_record(kname, N, a.model) or {}
        traffic = int(pmc["hbm_bytes_per_frame"] * N) if "hbm_bytes_per_frame" in pmc else None
        line = {
            "metric": METRIC if not stub else "LAUNCHER SELF-TEST (stub batch, no GPU work)",
            "value": round(frames_per_rep / med, 1), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * med / K, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int8 weights x u8 activations (i32 accumulate) + f32/f64 DSP",
            "data": "synthetic" + (", fed from and returned to pinned host memory over PCIe inside the timed region" if a.host_io else ""),
            "value_min": round(frames_per_rep / max(times), 1), "value_max": round(frames_per_rep / min(times), 1),
            "repeats": R,
            # each rank's own rate (its frames / the median of ITS repetition times): the aggregate is paced by the slowest
            "value_by_rank": [round(N * K / t, 1) for t in rank_medians],
            "config": {"workload": workload_name(a, N), "streams_per_gpu": N, "frames_per_step": N * world,
                       "nn_path": a.nn, "model": a.model, "outputs_sane": sane,
                       "stream_ids_rank0": [mine.start, mine.stop], "stream_ids_by_rank": shards},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(ach, 2), "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": round(ach * 1e9 / HBM_PEAK, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "launch_ms": round(dom_ms, 4),
                         "kernel_ms": {k: round(kms[k], 4) for k in kinds},
                         "launches_per_step": n_launch,
                         "note": "dominant kernel = longest single launch by HIP-event time inside the timed region (kernel_ms adds up "
                                 "the launches of a kind; overlapping kernels of neighbouring frames stretch each other's durations); "
                                 "it is LDS/issue-bound, not HBM-bound: see roofline_lds and roofline_valu"},
            # whole step against HBM: mandatory bytes of all four kernels / step time
            "roofline_step": {"bound": "hbm", "achieved": round(sum(per_launch.values()) / (med / K) / 1e9, 2),
                              "unit": "GB/s", "frac": round(sum(per_launch.values()) / (med / K) / HBM_PEAK, 5),
                              "algorithmic_bytes_per_step": sum(per_launch.values())},
            "weight_roofline": {"W_bytes_per_frame": W, "frac": round(frames_per_rep / med * W / (world * HBM_PEAK), 5),
                                "definition": "frames/s x W / (n_gpus x 8.0e12 B/s), north_star / SURVEY 8d; weights are "
                                              "L2-served, so this normalised figure may exceed 1"},
        }
        if kms_alone:
            da = max(kinds, key=lambda k: kms_alone[k] / n_launch[k])
            na = kernel_of(da, N, a.nn)
            pa = pmc_record(na, N, a.model) or {}
            da_ms, da_bytes = kms_alone[da] / n_launch[da], per_launch[da] // n_launch[da]
            aa = da_bytes / (da_ms * 1e-3) / 1e9
            line["roofline_standalone"] = {
                "bound": "hbm", "kernel": na, "achieved": round(aa, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(aa * 1e9 / HBM_PEAK, 5),
                "traffic": int(pa["hbm_bytes_per_frame"] * N) if "hbm_bytes_per_frame" in pa else None,
                "algorithmic_bytes_per_launch": da_bytes, "launch_ms": round(da_ms, 4),
                "kernel_ms": {k: round(kms_alone[k], 4) for k in kinds},
                "note": "same workload, every kernel on one stream (no overlap between kernels), outside the timed region"}
            wa = waves_per_launch(da, N)  # (of ONE launch, also for the layer-wise network)
            if "valu_per_wave" in pa:
                ti = pa["valu_per_wave"] * wa * valu_cost(pa.get("kernel", na)) / N_SIMD / clock_hz(pa)
                line["roofline_standalone"]["valu_issue_frac"] = round(ti / (da_ms * 1e-3), 4)
                line["roofline_standalone"]["valu_note"] = (f"{pa['valu_per_wave']} VALU instructions per wave (PMC) x "
                                                           f"{valu_cost(pa.get('kernel', na)):.2f} clk (instruction mix priced with profiles/r4_valu_issue.txt) "
                                                           f"over 1024 SIMDs at {clock_hz(pa) / 1e9:.2f} GHz")
                line["roofline_standalone"]["clock_ghz"] = round(clock_hz(pa) / 1e9, 3)
            if "lds_cycles_per_wave" in pa:
                tl = pa["lds_cycles_per_wave"] * wa / N_CU / clock_hz(pa)
                line["roofline_standalone"]["lds_port_frac"] = round(tl / (da_ms * 1e-3), 4)
        if "valu_per_wave" in pmc and dom_ms > 0:
            # VALU-issue bound of the dominant launch: instructions per wave (PMC) x the mean issue cost of this kernel's
            # instruction mix (valu_cost) x waves, spread over 1024 SIMDs at 2.4 GHz.  frac_f32_peak prices every instruction
            # at the plain-f32 rate instead (the guide's vector peak): a floor no mix with DPP / f64 / selects can reach.
            cpi = valu_cost(pmc.get("kernel", kname))
            t_issue = pmc["valu_per_wave"] * waves_per_launch(dom, N) * cpi / N_SIMD / clock_hz(pmc)
            line["roofline_valu"] = {"bound": "valu-issue", "kernel": kname, "valu_insts_per_wave": pmc["valu_per_wave"],
                                     "waves": waves_per_launch(dom, N), "clocks_per_inst": cpi,
                                     "clocks_per_inst_kind": valu_cost_kind(pmc.get("kernel", kname)),
                                     "clock_ghz": round(clock_hz(pmc) / 1e9, 3),
                                     "clock_source": "GRBM_GUI_ACTIVE / duration of the kernel's PMC pass" if pmc.get("clock_ghz") else "nominal maximum (no measurement in profiles/)",
                                     "issue_bound_ms": round(1e3 * t_issue, 4),
                                     "frac": round(t_issue / (dom_ms * 1e-3), 4),
                                     "frac_f32_peak": round(t_issue * VALU_FLOOR / cpi / (dom_ms * 1e-3), 4),
                                     "peak": "1024 SIMDs x clock_ghz / clocks_per_inst wave64 instructions/s; clocks_per_inst = this kernel's "
                                             "instruction mix priced with the per-instruction costs measured in profiles/r4_valu_issue.txt "
                                             "(frac_f32_peak: every instruction at 2.26 clk, the measured plain-f32 rate)",
                                     "source": pmc.get("source", "profiles/")}
        if "lds_cycles_per_wave" in pmc and dom_ms > 0:
            # LDS-port bound: SQ_LDS_IDX_ACTIVE cycles per wave (PMC: cycles the CU's one LDS pipe is busy for this wave,
            # bank-conflict replays included) x waves / 256 CUs / 2.4 GHz
            t_lds = pmc["lds_cycles_per_wave"] * waves_per_launch(dom, N) / N_CU / clock_hz(pmc)
            line["roofline_lds"] = {"bound": "lds-port", "kernel": kname, "lds_cycles_per_wave": pmc["lds_cycles_per_wave"],
                                    "waves": waves_per_launch(dom, N), "port_bound_ms": round(1e3 * t_lds, 4),
                                    "frac": round(t_lds / (dom_ms * 1e-3), 4),
                                    "clock_ghz": round(clock_hz(pmc) / 1e9, 3),
                                    "peak": "one LDS pipe per CU: 256 CUs x clock_ghz port-cycles/s", "source": pmc.get("source", "profiles/")}
        try:  # the whole step against VALU issue: every kernel of it is issue-bound to first order
            vi = step_valu_issue_ms(N, a.model, a.nn)
            if vi and med > 0:
                line["roofline_step_valu"] = {"bound": "valu-issue", "issue_bound_ms": round(vi, 4),
                                              "frac": round(vi / (1e3 * med / K), 4),
                                              "definition": "sum over the step's kernels of waves x VALU instructions per wave (PMC passes under "
                                                            "profiles/) x that kernel's mix-priced clocks per instruction (profiles/valu_mix.json, "
                                                            "profiles/r4_valu_issue.txt), over 1024 SIMDs at each kernel's measured clock, divided by ms_per_step"}
        except Exception:
            pass
        if not stub and a.nn == "mfma":
            # north_star's condition on the MFMA path ("rocprof shows real MFMA utilisation"): SQ_VALU_MFMA_BUSY_CYCLES of each network
            # kernel over (1,024 SIMDs x the launch's GPU-active cycles), from the PMC passes kept under profiles/
            mf = {}
            for kn in sorted(set(NN_LAYER_KERNELS if N >= NN_LAYERS_MIN_STREAMS else ("rn_nn_mfma_kernel",))):
                rec = pmc_record(kn, N, a.model) or {}
                if "mfma_busy_frac" in rec:
                    mf[kn] = {"busy_frac": rec["mfma_busy_frac"], "mfma_per_wave": rec.get("mfma_per_wave"), "source": rec.get("source")}
            if mf:
                line["mfma_utilisation"] = {"definition": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) per launch, one-stream schedule",
                                            "kernels": mf}
        line["devices_by_rank"] = devices
        buses = [d.get("pci_bus") for d in devices if d.get("pci_bus")]
        if len(buses) != len(set(buses)):
            line["devices_shared_between_ranks"] = True  # (not a scaling number: some ranks ran on the same GPU)
        if stub:
            line["stub"] = True
        if share and world > 1:
            line["shared_device"] = True
        if world > 1:
            line["collective"] = collective
            if collective_error:
                line["collective_error"] = collective_error
        if not stub and not a.no_parity:
            try:
                line["parity"] = parity_leg(capi, torch, batch, blob, d_in_f32, min(cap, 12), a.s16)
            except Exception as e:  # the checker failing to run is reported, never hidden
                line["parity"] = {"bit_identical": None, "error": repr(e)[:300]}
        if world == 1 and not a.no_cpu_baseline and not stub:
            cb = cpu_baseline(blob)
            if cb:
                line["cpu_baseline"] = cb
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=25, help="repetitions of the K-step timed region (median reported)")
    ap.add_argument("--streams", type=int, default=65536, help="concurrent streams PER GPU (configs[2]: 65536)")
    ap.add_argument("--model", choices=["default", "little"], default="default")
    ap.add_argument("--nn", choices=["vector", "mfma"], default=os.environ.get("RNNOISE_AMD_NN", "mfma"),
                    help="network path: batched MFMA (default) or the v_dot4 vector path; identical bits")
    ap.add_argument("--host-io", action="store_true", help="feed host buffers through rnnoise_batch_process (PCIe-inclusive)")
    ap.add_argument("--s16", action="store_true", help="int16 PCM at both ends (rnnoise_batch_process[_device]_s16)")
    ap.add_argument("--frames-per-call", type=int, default=0,
                    help="frames per library call (0 = as many as the step count allows; 1 = the cadence of a real-time server)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--smoke", action="store_true",
                    help="a two-step, one-repetition run of the same path (launcher, collectives, sharding, parity leg): what to run first on a new node")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)  # launcher self-test on CPU (gloo)
    a = ap.parse_args(argv)
    if a.smoke:
        a.steps, a.warmup, a.repeats, a.no_cpu_baseline = 2, 1, 1, True
    return a


def main():
    a = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and a.gpus > 1:
        sys.exit(relaunch_under_torchrun(a))
    line = bench_rank(a)
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
