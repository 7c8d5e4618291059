#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4j}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$R/rnnoise_amd -lpthread
BLOB=oracle/_ref/default.blob
for t in 1 16; do RNNOISE_AMD_FRAME_TAPS=1 timeout 120 /tmp/configs0_mt $BLOB $t 3000 2>&1 | sed "s/^/fused=1 /" | tee -a "$O/comb.txt"; done
RNNOISE_AMD_FUSED=0 timeout 120 /tmp/configs0_mt $BLOB 1 3000 2>&1 | sed "s/^/fused=0 /" | tee -a "$O/comb.txt"
cd /tmp; python "$R/tools/serial_times.py" 1 2>&1 | grep "N=" | tee -a "$O/comb.txt"
