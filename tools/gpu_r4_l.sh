#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4l}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py -m gpu -x -q 2>&1 | tail -2 | tee -a "$O/pytest.txt"
cd /tmp
for i in 1 2; do
  for x in 0 16384; do RNNOISE_AMD_K1_EXPERIMENT=$x python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/experiment=$x /" | tee -a "$O/serial.txt"; done
  RNNOISE_AMD_K1_SPREAD=0 python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/spread=0 /" | tee -a "$O/serial.txt"
done
