#!/bin/bash
# one stream per analysis workgroup against four (RNNOISE_AMD_K1_SPW) over the batch sizes around the switch, one gpurun call
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-spw}
mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
for i in 1 2; do
  for w in 1 4; do
    RNNOISE_AMD_K1_SPW=$w python "$R/tools/serial_times.py" 1024 2048 3072 4096 6144 8192 2>&1 | grep "N=" | sed "s/^/spw=$w /" | tee -a "$O/spw.txt"
  done
done
for w in 1 4; do
  RNNOISE_AMD_K1_SPW=$w python "$R/bench.py" --no-cpu-baseline --no-parity --streams 4096 --steps 50 --warmup 10 --frames-per-call 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spw=$w fpc1 4096', d['value'], d['ms_per_step'])" | tee -a "$O/spw.txt"
  RNNOISE_AMD_K1_SPW=$w python "$R/bench.py" --no-cpu-baseline --no-parity --streams 4096 --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spw=$w pipelined 4096', d['value'], d['ms_per_step'])" | tee -a "$O/spw.txt"
done
echo done
