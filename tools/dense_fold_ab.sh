#!/bin/bash
# profiles/r6_dense_fold.txt: the layer-wise network with dense_out / vad_dense folded into the front and layer launches (instrumented
# library, RNNOISE_AMD_GRU_VARIANT=w8f) against the five-launch network -- parity first (one small case under a SHORT timeout: the fold's
# waves wait for each other's flags), then stand-alone kernel times and the shader-clock taps of a layer workgroup's wave 0.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-fold}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
export RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_instr.so  # (every form but w4 / w8 lives in the instrumented library)
RNNOISE_AMD_GRU_VARIANT=w8f RNNOISE_AMD_NN_LAYERS_MIN=0 RNNOISE_AMD_NN_ONE_MAX=0 timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_mfma_path_bit_exact and 65-2" 2>&1 | tail -3 | tee "$O/quick.txt"
grep -q " passed" "$O/quick.txt" || { echo "not green"; exit 1; }
RNNOISE_AMD_GRU_VARIANT=w8f RNNOISE_AMD_NN_LAYERS_MIN=0 RNNOISE_AMD_NN_ONE_MAX=0 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_blob_tools.py -m gpu -x -q -k "test_mfma_path_bit_exact or test_sparser_blob_on_ragged_batches or test_synthetic_models_on_gpu" 2>&1 | tail -3 | tee -a "$O/quick.txt"
RNNOISE_AMD_GRU_VARIANT=w8f timeout 400 python -m pytest tests/test_gpu_at_size.py -m gpu -x -q -k "65536_stream_batch or ragged_40037" 2>&1 | tail -3 | tee -a "$O/quick.txt"
for v in "" w8 w8f; do
  echo "variant=${v:-default(w4)}: $(RNNOISE_AMD_GRU_VARIANT=$v timeout 120 python tools/serial_times.py 65536 2>&1 | tail -1)" | tee -a "$O/quick.txt"
done
for v in w8 w8f; do
  echo "== $v" | tee -a "$O/quick.txt"
  RNNOISE_AMD_GRU_VARIANT=$v timeout 200 python tools/k1_cycles.py 65536 --nn --layers 2>&1 | grep -A20 "N=65536: MFMA" | tail -10 | grep -v "^---\|load  " | tee -a "$O/quick.txt"
done
