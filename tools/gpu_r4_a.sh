#!/bin/bash
# round 4, GPU call A: full GPU tests -> stand-alone kernel times with the narrow phases spread / on wave 0 -> bench line
# -> K1 by section.   usage: tools/gpu_r4_a.sh TAG
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r4a}
O=$R/gpurun_out/$T
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest.txt"
tail -5 "$O/pytest.txt"
cd /tmp
for sp in 1 0; do
  RNNOISE_AMD_K1_SPREAD=$sp python "$R/tools/serial_times.py" 16384 65536 2>&1 | grep "N=" | sed "s/^/spread=$sp /" | tee -a "$O/serial_times.txt"
done
python "$R/bench.py" --no-cpu-baseline --repeats 9 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_65536.json"
RNNOISE_AMD_K1_SPREAD=0 python "$R/bench.py" --no-cpu-baseline --repeats 9 > "$O/b0.log" 2>&1; grep '^{' "$O/b0.log" | tail -1 > "$O/bench_65536_spread0.json"
python -c "
import json
for f in ('bench_65536.json','bench_65536_spread0.json'):
    d=json.load(open('$O/'+f)); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity'))"
bash "$R/tools/k1_prefix.sh" "$T/prefix" 65536 > /dev/null 2>&1
cat "$O/prefix/k1_prefix.txt"
