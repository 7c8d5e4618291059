#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X rnnoise_process_frame() path.

Metric (BASELINE.json): 10 ms frames/s (48 kHz mono) summed over N concurrent streams, and
the fraction of the HBM roofline.  A "step" is one pass of the hot path over one batch: every
stream of the batch advances by one 480-sample frame (analysis -> network -> synthesis).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--nn vector|mfma]

Workload at N=1: BASELINE.json configs[1] -- 4096 concurrent streams on one MI355X, default
architecture, int8 model -- run on the faster of the two bit-identical network paths (batched
MFMA; `--nn vector` selects the v_dot4 path configs[1] names, `--streams 65536` is configs[2]).  With --gpus N every rank owns its own S streams (independent
streams shard trivially, SURVEY 8e: "weak" scaling, no data-path collective); the only
collectives are the barrier and the max/sum over ranks of (elapsed, frames).

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
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)
FRAME = 480

# algorithmic HBM bytes per stream-frame of each kernel (DESIGN.md "kernels"); W is added to
# the network kernel at run time from the model
# analysis = K0 (in 1920 r, ring slot 1920 w, hp state 16) + K1 (ring: 6912 downsample + 2 x 3840 windows r,
# X re-read 3200 r; X,P 7696 + E 384 + features 260 + flags 12 w)
ANALYSIS_BYTES = (1920 + 1920 + 16) + (6912 + 3840 + 3840 + 3200) + (7696 + 384 + 260 + 12)
SYNTHESIS_BYTES = 3848 + 3848 + 384 + 128 + 128 + 256 + 1920 + 1920 + 1920
NETWORK_STATE_BYTES = 260 + 2 * (520 + 1024 + 4608) + 128 + 4


def load_blob() -> bytes:
    with open(os.path.join(ROOT, "tests", "golden", "default.blob.xz"), "rb") as f:
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


def measured_traffic(kernel: str, n_streams: int):
    """HBM bytes per launch from the committed PMC passes (profiles/r1_traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 x2 read correction; measured at 4096 and 65,536 streams),
    per-frame figure of the nearer measurement scaled to this batch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as f:
            sets = json.load(f)["by_streams"]
        k = sets[min(sets, key=lambda n: (abs(math.log2(int(n) / n_streams)), -int(n)))]  # nearest in octaves
        kernel = kernel.replace("_lean", "")  # the 80-VGPR build is listed under the plain name
        if kernel == "rn_analysis_kernel":
            per = k["rn_analysis_kernel"]["hbm_bytes_per_frame"] + k["rn_hp_kernel"]["hbm_bytes_per_frame"]
        else:
            per = k[kernel]["hbm_bytes_per_frame"]
        return int(per * n_streams)
    except Exception:
        return None


def cpu_baseline(blob: bytes):
    """Reference (or port) on the host cores, bounded to ~12 s; see oracle/cpu_bench.c."""
    import numpy as np
    from rnnoise_amd import synth
    ref = os.path.join(ROOT, "oracle", "_ref", "cpu_bench_ref")
    port = os.path.join(ROOT, "oracle", "cpu_bench_port")
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as td:
        bp, pp = os.path.join(td, "m.blob"), os.path.join(td, "pcm.s16")
        open(bp, "wb").write(blob)
        np.concatenate([synth.stream_pcm(s, 200) for s in range(8)]).tofile(pp)
        for exe, kind in ((ref, "reference"), (port, "port")):
            if not os.path.exists(exe):
                if kind == "port":
                    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "cpu_bench_port"],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                if not os.path.exists(exe):
                    continue
            try:
                r = subprocess.run([exe, bp, pp, str(cores), "12"], capture_output=True, text=True, timeout=120)
                j = json.loads(r.stdout.strip().splitlines()[-1])
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
            return {"value": round(j["frames_per_s"], 1), "unit": "frames/s", "cores": j["threads"], "kind": kind,
                    "sample": f"{j['frames']} frames in {j['seconds']:.1f} s: {cores} threads x 1 stream each, "
                              f"200-frame synthetic PCM looped in memory, same blob; host CPU {cpu}"}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=4096, help="concurrent streams PER GPU (configs[1]: 4096)")
    ap.add_argument("--nn", choices=["vector", "mfma"], default=os.environ.get("RNNOISE_AMD_NN", "mfma"),
                    help="network path: batched MFMA (default) or the v_dot4 vector path; identical bits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    from rnnoise_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch multi-GPU runs with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    blob = load_blob()
    model = capi.Model(blob)
    W = model.weight_bytes
    N, K, Wm = a.streams, a.steps, a.warmup
    batch = capi.Batch(model, N, device=local_rank)
    batch.set_nn_path(1 if a.nn == "mfma" else 0)

    # inputs resident in HBM; cycle through at most `cap` distinct frames if K+W is large
    cap = max(8, min(K + Wm, (3 << 30) // (N * FRAME * 4)))
    d_in = synth_pcm_torch(torch, N, cap, dev, seed_base=rank * N)
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty((cap, N), device=dev)
    d_gains = torch.empty((cap, N, 32), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    esz = N * FRAME * 4

    def run(first: int, count: int):
        f = first
        left = count
        while left > 0:
            k = f % cap
            n = min(left, cap - k)
            batch.process_device(d_out.data_ptr() + k * esz, d_in.data_ptr() + k * esz, d_vad.data_ptr() + k * N * 4,
                                 d_gains.data_ptr() + k * N * 128, n, stream)
            f += n
            left -= n

    run(0, Wm)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    batch.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(Wm, K)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = batch.kernel_ms()
    batch.enable_timing(False)

    from rnnoise_amd.dist import aggregate_throughput
    frames, dt = aggregate_throughput(float(N * K), dt, dist, dev)
    value = frames / dt

    if rank == 0:
        sane = bool(torch.isfinite(d_out).all().item()) and float(d_vad.max().item()) > 0.0
        # algorithmic HBM bytes per launch: per-stream traffic x streams; the weights are read from HBM at most
        # once per launch, whatever the batch (every later tile finds them in L2 / Infinity Cache) -- the
        # north_star's per-frame W figure is reported separately as weight_roofline
        per_launch = {"analysis": ANALYSIS_BYTES * N, "network": W + NETWORK_STATE_BYTES * N, "synthesis": SYNTHESIS_BYTES * N}
        dom = max(("analysis", "network", "synthesis"), key=lambda k: kms[k])
        ach = per_launch[dom] / (kms[dom] * 1e-3) / 1e9 if kms[dom] > 0 else 0.0
        kname = f"rn_{dom}_kernel" if dom != "network" else f"rn_nn_{a.nn}_kernel"
        if dom == "analysis" and 3072 <= N < 24576 and os.environ.get("RNNOISE_AMD_K1_LEAN", "1") != "0":
            kname = "rn_analysis_lean_kernel"  # same code held to 80 VGPRs (dsp_kernels.hip: RN_K1_LEAN_MIN/MAX_STREAMS)
        line = {
            "metric": "10ms frames/sec (48kHz mono) at N concurrent streams; % HBM roofline",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": a.gpus, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int8 weights x u8 activations (i32 accumulate) + f32/f64 DSP",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {N} concurrent streams per GPU, default architecture "
                                   f"(conv 65x3->128->384, 3xGRU(384) block-sparse int8, density 1/3), synthetic "
                                   f"exporter-made model, network path = {a.nn}",
                       "streams_per_gpu": N, "frames_per_step": N * a.gpus, "nn_path": a.nn,
                       "outputs_sane": sane},
            "roofline": {"bound": "hbm", "kernel": kname + (" (+rn_hp_kernel)" if dom == "analysis" else ""),
                         "achieved": round(ach, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(ach * 1e9 / HBM_PEAK, 5), "traffic": measured_traffic(kname, N),
                         "algorithmic_bytes_per_launch": per_launch[dom],
                         "kernel_ms": {k: round(kms[k], 4) for k in ("analysis", "network", "synthesis")}},
            "weight_roofline": {"W_bytes_per_frame": W, "frac": round(value * W / (a.gpus * HBM_PEAK), 5),
                                "definition": "frames/s x W / (n_gpus x 8.0e12 B/s), north_star / SURVEY 8d"},
        }
        if a.gpus == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(blob)
            if cb:
                line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
