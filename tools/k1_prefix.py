#!/usr/bin/env python3
"""Driver for the K1 prefix profile: a few frames of an N-stream batch through the INSTRUMENTED library on one stream, with
$RNNOISE_AMD_K1_STOP deciding where rn_analysis_kernel's workgroups leave (dsp_kernels.hip: K1_STOP).  Run under rocprofv3
by tools/k1_prefix.sh; prints the mean duration of the analysis kernel itself as well."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rnnoise_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
capi.instrumented().__enter__()
dev = torch.device("cuda:0")
model = capi.Model(bench.load_blob())
b = capi.Batch(model, N)
b.set_schedule(9)
d_in = bench.synth_pcm_torch(torch, N, frames + 2, dev, seed_base=0)
d_out = torch.empty_like(d_in)
d_vad = torch.empty((frames + 2, N), device=dev)
st = torch.cuda.current_stream().cuda_stream
b.process_device(d_out.data_ptr(), d_in.data_ptr(), d_vad.data_ptr(), 0, 2, st)
torch.cuda.synchronize()
b.enable_timing(True)
b.process_device(d_out[2].data_ptr(), d_in[2].data_ptr(), d_vad[2].data_ptr(), 0, frames, st)
torch.cuda.synchronize()
print(f"stop={os.environ.get('RNNOISE_AMD_K1_STOP', '0')} N={N} analysis_ms={b.kernel_ms()['analysis']:.4f}")
