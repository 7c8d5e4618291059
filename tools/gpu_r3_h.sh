#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "blob_tools or drop_in or dropin or free_running or invariance or fuzz or mfma_path or chunking or extreme or golden" 2>&1 | tail -3
python tools/nn_one_taps.py 1 2>&1 | grep -v amdgpu | tail -8
cd /tmp
python "$R/tools/serial_times.py" 1 64 256 2>&1 | grep "N="
python "$R/tools/configs0.py" 2>&1 | grep "pooled"
true
