#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3b
mkdir -p "$O"; export TMPDIR=/tmp
timeout 600 "$R/rnnoise_amd/csrc/build/valu_issue" > "$O/valu_issue.txt" 2>&1
cd /tmp
python "$R/tools/k1_cycles.py" 65536 2>&1 | grep -v amdgpu.ids > "$O/section_taps_65536.txt"
