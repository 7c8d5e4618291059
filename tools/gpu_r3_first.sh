#!/bin/bash
# round 3, first call on the GPU box: host rcpps analysis + table, VALU issue table, baseline bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p "$O"; export TMPDIR=/tmp
grep -m1 "model name" /proc/cpuinfo > "$O/cpu.txt"; nproc >> "$O/cpu.txt"
( cd "$R/oracle/_ref" && ./rcp_capture --analyze /tmp/rcp_full.bin > "$O/rcp_analyze.txt" 2>&1; ./rcp_capture "$O/rcp_lut_host.h" >> "$O/rcp_analyze.txt" 2>&1; echo "rc=$?" >> "$O/rcp_analyze.txt" )
python - <<PY
import lzma
d=open('/tmp/rcp_full.bin','rb').read()
open('$O/rcp_full_1_2.bin.xz','wb').write(lzma.compress(d, preset=6))
PY
timeout 300 "$R/rnnoise_amd/csrc/build/valu_issue" > "$O/valu_issue.txt" 2>&1
cd /tmp
timeout 600 python "$R/bench.py" --no-cpu-baseline > "$O/bench_65536.log" 2>&1; grep '^{' "$O/bench_65536.log" | tail -1 > "$O/bench_65536.json"
ls -la "$O"
