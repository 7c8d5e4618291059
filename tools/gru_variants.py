#!/usr/bin/env python3
"""A/B of the GRU layer kernel's variants ($RNNOISE_AMD_GRU_VARIANT, nn_layers.hip) on a GPU, one process per variant:
  parity   a RAGGED batch (20,000 streams = 625 replicas of a 32-stream block: 313 groups of 64, the last one half empty, so the
           persistent variants' workgroups take one or two groups) for 10 frames as calls of 3 + 1 + 6 on the default schedule,
           replicas compared with each other on the GPU and the first block with the oracle (bit for bit: PCM, gains, VAD);
  time     65,536 streams: the network's launches stand-alone (HIP events, one stream; mean of 20 steps, best of 3) and the
           pipelined throughput of 8-frame calls (median of 7).

usage: tools/gru_variants.py [variant ...]      default: every variant of the product build worth comparing
       tools/gru_variants.py --worker <variant> <oracle.npz>     (internal)
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DEFAULT = ["w8", "w4", "p", "v3", "v3nobd", "v3np", "v3mprio"]
N_PAR, CALLS = 20000, (3, 1, 6)


def base_pcm():
    from rnnoise_amd import synth
    T = sum(CALLS)
    base = synth.batch_pcm(range(32), T)
    base[:3, 21] = 0
    base[T - 5:T - 3, 24] = 0
    return base


def worker(variant, npz):
    import torch

    import bench
    from rnnoise_amd import capi
    dev = torch.device("cuda:0")
    want = np.load(npz)
    model = capi.Model(bench.load_blob())
    # ---- parity ----
    base, T, N = want["base"], sum(CALLS), N_PAR
    d_in = torch.from_numpy(base).to(dev).repeat(1, N // 32, 1).contiguous()
    d_out, d_vad, d_gains = torch.empty_like(d_in), torch.empty((T, N), device=dev), torch.empty((T, N, 32), device=dev)
    b = capi.Batch(model, N)
    b.set_nn_path(2)  # the layer-wise network (the default from 16,384 streams up)
    st = torch.cuda.current_stream().cuda_stream
    f = 0
    for n in CALLS:
        b.process_device(d_out[f].data_ptr(), d_in[f].data_ptr(), d_vad[f].data_ptr(), d_gains[f].data_ptr(), n, st)
        f += n
    torch.cuda.synchronize()
    same = all(bool((t.view(torch.int32).reshape(T, N // 32, 32 * w) == t.view(torch.int32).reshape(T, N // 32, 32 * w)[:, :1]).all().item())
               for t, w in ((d_out, 480), (d_gains, 32), (d_vad, 1)))
    eq = all(np.array_equal(t[:, :32].cpu().numpy().view(np.uint32), want[k].view(np.uint32))
             for t, k in ((d_out, "out"), (d_gains, "gains"), (d_vad, "vad")))
    tail = all(np.array_equal(t[:, N - 32:].cpu().numpy().view(np.uint32), want[k].view(np.uint32))
               for t, k in ((d_out, "out"), (d_gains, "gains"), (d_vad, "vad")))
    b.close()
    del d_in, d_out, d_vad, d_gains
    # ---- time ----
    N, cap = 65536, 8
    d_in = bench.synth_pcm_torch(torch, N, cap, dev, seed_base=0)
    d_out, d_vad = torch.empty_like(d_in), torch.empty((cap, N), device=dev)
    esz = N * 480 * 4
    b = capi.Batch(model, N)

    def one(k):
        b.process_device(d_out.data_ptr() + k * esz, d_in.data_ptr() + k * esz, d_vad.data_ptr() + k * N * 4, 0, 1, st)

    for k in range(6):
        one(k)
    torch.cuda.synchronize()
    k2 = []
    for _ in range(3):
        b.enable_timing(True)
        for k in range(20):
            one(k % cap)
        torch.cuda.synchronize()
        k2.append(b.kernel_ms()["network"])
        b.enable_timing(False)
    thr = []
    for _ in range(7):
        b.process_device(d_out.data_ptr(), d_in.data_ptr(), d_vad.data_ptr(), 0, cap, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            b.process_device(d_out.data_ptr(), d_in.data_ptr(), d_vad.data_ptr(), 0, cap, st)
        torch.cuda.synchronize()
        thr.append(N * cap * 3 / (time.perf_counter() - t0) / 1e6)
    b.close()
    print(f"{variant:8s} replicas equal {same}  first block == oracle {eq}  last block == oracle {tail} | K2 stand-alone {min(k2):.4f} ms "
          f"(runs {' '.join('%.4f' % x for x in k2)}) | pipelined {sorted(thr)[len(thr) // 2]:.2f} M frames/s (min {min(thr):.2f} max {max(thr):.2f})",
          flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(sys.argv[2], sys.argv[3])
    import bench
    from oracle import binding
    from test_gpu_parity import oracle_run
    binding.set_rcp_profile("host")  # (the product's default profile: this machine's rcpps)
    variants = sys.argv[1:] or DEFAULT
    base = base_pcm()
    want = oracle_run(bench.load_blob(), base, collect_state=False)
    print(f"# tools/gru_variants.py: parity on {N_PAR} streams (ragged last group), calls of {CALLS}; times at 65,536 streams; "
          f"grid override RNNOISE_AMD_GRU_GRID={os.environ.get('RNNOISE_AMD_GRU_GRID', '-')}", flush=True)
    with tempfile.TemporaryDirectory() as td:
        npz = os.path.join(td, "want.npz")
        np.savez(npz, base=base, out=want["out"], gains=want["gains"], vad=want["vad"])
        for v in variants:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", v, npz], # (every form but w4 / w8 lives in the instrumented library only: the whole table is measured on that one)
                               env=dict(os.environ, RNNOISE_AMD_GRU_VARIANT=v, RNNOISE_AMD_LIB=os.path.join(ROOT, "rnnoise_amd", "librnnoise_amd_instr.so")),
                               capture_output=True, text=True, timeout=600)
            out = [ln for ln in r.stdout.splitlines() if ln.startswith(v)]
            print(out[-1] if out else f"{v:8s} FAILED rc={r.returncode}: {(r.stderr or r.stdout)[-600:]}", flush=True)


if __name__ == "__main__":
    main()
