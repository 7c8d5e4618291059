#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4h}
mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
RNNOISE_AMD_GRU_W=4 python $R/tools/diag_replicas.py 32768 little 10 2>&1 | grep rep | sed "s/^/W=4 little /" | tee -a "$O/res.txt"
RNNOISE_AMD_GRU_W=4 python $R/tools/diag_replicas.py 32768 default 6 2>&1 | grep rep | sed "s/^/W=4 default /" | tee -a "$O/res.txt"
RNNOISE_AMD_GRU_W=8 python $R/tools/diag_replicas.py 32768 little 6 2>&1 | grep rep | sed "s/^/W=8 little /" | tee -a "$O/res.txt"
