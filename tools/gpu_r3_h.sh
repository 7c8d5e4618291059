#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; export TMPDIR=/tmp
python -c "
import lzma; open('/tmp/w.blob','wb').write(lzma.decompress(open('tests/golden/default.blob.xz','rb').read()))"
gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$R/rnnoise_amd -lpthread
for t in 1 2 4 8 16; do /tmp/configs0_mt /tmp/w.blob $t 2000 2>&1 | grep configs; done
GPU_MAX_HW_QUEUES=8 /tmp/configs0_mt /tmp/w.blob 8 2000 2>&1 | grep configs
true
