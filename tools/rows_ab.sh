#!/bin/bash
# (RNNOISE_AMD_ROWS_K1 is an A/B switch of the instrumented library: the harness links against that one)
# one-frame API: analysis as one workgroup of four waves per row (default) against one wave per row (RNNOISE_AMD_ROWS_K1=1), one gpurun call
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-rows}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"; timeout 900 python -m pytest tests/test_dropin_gpu.py tests/test_demo_dropin.py -m gpu -x -q 2>&1 | tail -3 | tee -a "$O/pytest.txt"
( cd "$R" && gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd_instr.so -Wl,-rpath,$R/rnnoise_amd -lpthread )
python -c "import lzma;open('/tmp/default.blob','wb').write(lzma.decompress(open('$R/tests/golden/default.blob.xz','rb').read()))"
cd /tmp
for i in 1 2; do
  for v in 0 1; do
    for t in 1 4 16 64; do RNNOISE_AMD_ROWS_K1=$v timeout 120 /tmp/configs0_mt /tmp/default.blob $t 3000 2>&1 | sed "s/^/one_wave=$v  /" | tee -a "$O/cthreads.txt"; done
  done
done
echo done
