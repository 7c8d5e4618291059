#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3h; mkdir -p $O
cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.txt"
grep -v "^  \|^$" "$O/pytest_gpu.txt" | tail -5
cd /tmp
python "$R/tools/serial_times.py" 16384 65536 2>&1 | grep "N="
python "$R/bench.py" --no-cpu-baseline 2>&1 | grep '^{' > $O/bench.json; cut -c80-200 $O/bench.json; grep -o '"parity": {[^}]*}' $O/bench.json; grep -o '"roofline_lds": {[^}]*}' $O/bench.json
true
