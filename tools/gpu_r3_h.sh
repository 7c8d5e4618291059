#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3h; mkdir -p $O
cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.txt"
grep -v "^  \|^$" "$O/pytest_gpu.txt" | tail -5
cd /tmp
python "$R/tools/serial_times.py" 1 64 1024 4096 16384 65536 2>&1 | grep "N=" | tee $O/serial_times.txt
python "$R/tools/serial_times.py" --vector 256 512 2>&1 | grep "N="
python "$R/tools/configs0.py" 2>&1 | grep "configs" | tee $O/configs0.txt
true
