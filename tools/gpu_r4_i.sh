#!/bin/bash
# round 4, GPU call I: full GPU tests twice (stability after the GRU revert) + combiner knobs from C threads
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4i}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee -a "$O/pytest.txt"; done
gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$R/rnnoise_amd -lpthread
BLOB=oracle/_ref/default.blob
r() { local tag=$1; shift; for t in 4 16 64; do env "$@" timeout 120 /tmp/configs0_mt $BLOB $t 3000 2>&1 | sed "s/^/$tag /" | tee -a "$O/comb.txt"; done; }
r "default        " X=1
r "linger10       " RNNOISE_AMD_COMBINE_LINGER_US=10
r "linger25       " RNNOISE_AMD_COMBINE_LINGER_US=25
r "linger50       " RNNOISE_AMD_COMBINE_LINGER_US=50
r "hwq8 streams3  " GPU_MAX_HW_QUEUES=8
r "hwq8 streams6  " GPU_MAX_HW_QUEUES=8 RNNOISE_AMD_COMBINE_STREAMS=6
r "hwq8 streams8  " GPU_MAX_HW_QUEUES=8 RNNOISE_AMD_COMBINE_STREAMS=8
r "hwq8 s6 ling25 " GPU_MAX_HW_QUEUES=8 RNNOISE_AMD_COMBINE_STREAMS=6 RNNOISE_AMD_COMBINE_LINGER_US=25
