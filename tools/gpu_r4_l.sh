#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4l}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee -a "$O/pytest.txt"
cd /tmp
for i in 1 2; do
  for x in 0 2048; do RNNOISE_AMD_K1_EXPERIMENT=$x python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/experiment=$x /" | tee -a "$O/serial.txt"; done
  RNNOISE_AMD_K1_SPREAD=0 python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/spread=0 /" | tee -a "$O/serial.txt"
done
bash $R/tools/k1_narrow.sh "${1:-r4l}/narrow" > /dev/null 2>&1; cat "$O/narrow/narrow.txt"
python "$R/bench.py" --no-cpu-baseline --repeats 9 --steps 20 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity',{}).get('bit_identical'))" | tee -a "$O/serial.txt"
