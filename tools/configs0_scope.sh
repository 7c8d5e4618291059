#!/bin/bash
# Runs ON THE GPU BOX: the reference's own call (rnnoise_process_frame) from plain C threads through the combiner -- throughput AND the CPU
# time it costs (tools/configs0_mt.c) -- for thread counts up to 64, state counts up to 1,024 (more states than a pool's 256 rows, more
# callers than a launch group's 64 entries), with the followers' pre-sleep (round 5 default) and spinning from the start (round 4).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
( cd "$R" && gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$R/rnnoise_amd -lpthread )
python -c "import lzma;open('/tmp/default.blob','wb').write(lzma.decompress(open('$R/tests/golden/default.blob.xz','rb').read()))"
echo "# tools/configs0_scope.sh: $(nproc) CPUs visible, cgroup quota $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "# followers sleep through most of a group's time, then spin (RNNOISE_AMD_COMBINE_WAKE_EARLY_US=40, the default)"
for t in 1 4 16 32 64; do timeout 120 /tmp/configs0_mt /tmp/default.blob $t 3000 2>&1; done
echo "# followers spin from the start (RNNOISE_AMD_COMBINE_WAKE_EARLY_US=0: round 4)"
for t in 4 16 32 64; do RNNOISE_AMD_COMBINE_WAKE_EARLY_US=0 timeout 120 /tmp/configs0_mt /tmp/default.blob $t 3000 2>&1; done
echo "# wake 70 us early"
for t in 16 64; do RNNOISE_AMD_COMBINE_WAKE_EARLY_US=70 timeout 120 /tmp/configs0_mt /tmp/default.blob $t 3000 2>&1; done
echo "# more states than threads: 64 threads over 128 / 256 / 1,024 states (one pool of 1,024 rows, the default)"
for s in 128 256 1024; do timeout 200 /tmp/configs0_mt /tmp/default.blob 64 $((60000 / s)) $s 2>&1; done
echo "# ... with pools of 256 rows (1, 1 and 4 pools)"
for s in 128 256 1024; do RNNOISE_AMD_POOL_ROWS=256 timeout 200 /tmp/configs0_mt /tmp/default.blob 64 $((60000 / s)) $s 2>&1; done
echo "# ... with pools of 64 rows (round 4's size: 2, 4 and 16 pools with three streams each)"
for s in 128 256 1024; do RNNOISE_AMD_POOL_ROWS=64 timeout 200 /tmp/configs0_mt /tmp/default.blob 64 $((60000 / s)) $s 2>&1; done
echo "# 128 threads over 256 states (a queue longer than a row list)"
timeout 200 /tmp/configs0_mt /tmp/default.blob 128 300 256 2>&1
echo "# a stream per state (RNNOISE_AMD_COMBINE=0: round 3)"
for t in 4 16; do RNNOISE_AMD_COMBINE=0 timeout 200 /tmp/configs0_mt /tmp/default.blob $t 2000 2>&1; done
