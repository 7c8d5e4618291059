#!/bin/bash
# the combiner's gather / linger windows against each other inside one gpurun call (tools/configs0_mt.c, 4 / 16 / 64 threads)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-knobs}
mkdir -p "$O"; export TMPDIR=/tmp
( cd "$R" && gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$R/rnnoise_amd -lpthread )
python -c "import lzma;open('/tmp/default.blob','wb').write(lzma.decompress(open('$R/tests/golden/default.blob.xz','rb').read()))"
cd /tmp
for i in 1 2; do
  for gl in "15 10" "10 10" "5 10" "15 5" "10 5" "5 5" "20 10" "10 0"; do
    set -- $gl
    for t in 2 4 16 64; do
      RNNOISE_AMD_COMBINE_GATHER_US=$1 RNNOISE_AMD_COMBINE_LINGER_US=$2 timeout 120 /tmp/configs0_mt /tmp/default.blob $t 3000 2>&1 | sed "s/^/gather=$1 linger=$2  /" | tee -a "$O/knobs.txt"
    done
  done
done
echo done
