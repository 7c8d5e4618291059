#!/bin/bash
# round 4, GPU call J: fused frame kernel -- drop-in tests, C threads fused / four launches
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4j}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_dropin_gpu.py tests/test_demo_dropin.py -m gpu -x -q 2>&1 | tail -3 | tee -a "$O/pytest.txt"
gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$R/rnnoise_amd -lpthread
BLOB=oracle/_ref/default.blob
for f in 1 0 1 0; do for t in 1 4 16 64; do RNNOISE_AMD_FUSED=$f timeout 120 /tmp/configs0_mt $BLOB $t 3000 2>&1 | sed "s/^/fused=$f /" | tee -a "$O/comb.txt"; done; done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee -a "$O/pytest.txt"
