#!/bin/bash
# round 4, GPU call E: GPU tests -> GRU activation batching A/B (kernel times, taps)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r4e}
O=$R/gpurun_out/$T
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest.txt"
tail -4 "$O/pytest.txt"
cd /tmp
for rep in 1 2; do
  python "$R/tools/serial_times.py" 16384 65536 2>&1 | grep "N=" | sed "s/^/default /" | tee -a "$O/serial_times.txt"
  RNNOISE_AMD_GRU_ACT=0 python "$R/tools/serial_times.py" 16384 65536 2>&1 | grep "N=" | sed "s/^/gru_act=0 /" | tee -a "$O/serial_times.txt"
done
python "$R/tools/k1_cycles.py" 65536 --nn --layers 2>&1 | grep -A8 "layer-wise GRU" | tee "$O/gru_taps.txt"
python "$R/bench.py" --no-cpu-baseline --repeats 9 --steps 20 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_65536.json"
python -c "
import json
d=json.load(open('$O/bench_65536.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity',{}).get('bit_identical'))"
