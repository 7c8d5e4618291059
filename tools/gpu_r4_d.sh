#!/bin/bash
# round 4, GPU call D: full GPU tests (new: two ranks on one device, host profile at 65,536, 10 s demo) -> GRU taps -> times -> bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r4d}
O=$R/gpurun_out/$T
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest.txt"
tail -8 "$O/pytest.txt"
cd /tmp
python "$R/tools/k1_cycles.py" 65536 --nn --layers 2>&1 | grep -v amdgpu.ids | tee "$O/section_taps_65536.txt"
python "$R/tools/serial_times.py" 1 65536 2>&1 | grep "N=" | tee -a "$O/serial_times.txt"
python "$R/bench.py" --no-cpu-baseline --repeats 9 --steps 20 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_65536.json"
python -c "
import json
d=json.load(open('$O/bench_65536.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity',{}).get('bit_identical'))"
