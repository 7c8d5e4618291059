#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "mfma_path or layerwise or at_size or 16384" 2>&1 | tail -3
cd /tmp
python "$R/tools/serial_times.py" 16384 65536 2>&1 | grep "N="
python "$R/bench.py" --no-cpu-baseline --no-parity 2>&1 | grep '^{' | cut -c80-220
true
