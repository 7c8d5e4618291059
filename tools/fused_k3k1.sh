#!/bin/bash
# VERDICT r5 Next #2, the experiment: synthesis(f-1) as the prologue of analysis(f) in one kernel (instrumented library,
# $RNNOISE_AMD_FUSE_K3K1=1) against the default three-stream pipeline -- parity first (the at-size checks against the oracle), then time,
# every line on the SAME library inside one gpurun call.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6c}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
export RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_instr.so
echo "== parity of the fused schedule (65,536 streams 5 + 1 + 8 frames; ragged 40,037; sparser blob 32,768): pytest on the instrumented library" | tee "$O/fused.txt"
RNNOISE_AMD_FUSE_K3K1=1 timeout 1200 python -m pytest tests/test_gpu_at_size.py -m gpu -x -q -k "65536_stream_batch or ragged_40037 or sparser_model_32768" 2>&1 | tail -4 | tee -a "$O/fused.txt"
last() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(f\"{d['value']/1e6:8.2f} M frames/s  {d['ms_per_step']:.4f} ms/step  (min {d['value_min']/1e6:.2f} max {d['value_max']/1e6:.2f})  parity {d.get('parity',{}).get('bit_identical')}  inside the pipeline: {d['roofline']['kernel_ms']}\")"; }
for i in 1 2 3; do
  echo "default (three streams: K1(f+1) beside K2(f) + K3(f)):  $(timeout 300 python bench.py --no-cpu-baseline --repeats 9 2>&1 | last)" | tee -a "$O/fused.txt"
  echo "fused   ([K3(f-1) . K1(f)] -> K2(f), one stream + K0):   $(RNNOISE_AMD_FUSE_K3K1=1 timeout 300 python bench.py --no-cpu-baseline --repeats 9 2>&1 | last)" | tee -a "$O/fused.txt"
  echo "one stream (K1 -> K2 -> K3, K0 aside = schedule 1):       $(RNNOISE_AMD_PIPE=1 timeout 300 python bench.py --no-cpu-baseline --repeats 9 2>&1 | last)" | tee -a "$O/fused.txt"
done
echo done
