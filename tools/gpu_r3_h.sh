#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3h
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.txt"
grep -v "^  \|^$" "$O/pytest_gpu.txt" | tail -12
cd /tmp
python "$R/tools/serial_times.py" 1 64 2>&1 | grep "N="
RNNOISE_AMD_NN_ONE_MAX=0 python "$R/tools/serial_times.py" 1 2>&1 | grep "N="
python "$R/bench.py" --no-cpu-baseline --no-parity --streams 4096 --nn vector --steps 50 --warmup 10 --repeats 5 2>&1 | grep '^{' | cut -c80-200
python "$R/tools/configs0.py" 2>&1 | grep configs
