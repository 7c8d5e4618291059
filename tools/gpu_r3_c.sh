#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
( cd oracle/_ref && ./rcp_capture AMD_ZEN5 "$O/rcp_profile_amd_zen5.h" > "$O/rcp_capture.txt" 2>&1; echo "rc=$?" >> "$O/rcp_capture.txt" )
cmp "$O/rcp_profile_amd_zen5.h" rnnoise_amd/csrc/rcp_profile_amd_zen5.h >> "$O/rcp_capture.txt" 2>&1 && echo "identical to the committed table" >> "$O/rcp_capture.txt"
python tests/golden/make_golden.py "$O/golden_amd_zen5" > "$O/make_golden.txt" 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -s > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.txt"
timeout 600 rnnoise_amd/csrc/build/valu_issue > "$O/valu_issue.txt" 2>&1
cd /tmp
python "$R/bench.py" --no-cpu-baseline > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_65536.json"
python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_4096.json"
python "$R/tools/serial_times.py" 4096 16384 65536 2>&1 | grep "N=" > "$O/serial_times.txt"
