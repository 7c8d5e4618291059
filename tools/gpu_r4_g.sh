#!/bin/bash
# round 4, GPU call G: GPU tests -> pipelined bench with the GRU layer kernel at 4 / 8 waves per workgroup (alternating) -> stand-alone times
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r4g}
O=$R/gpurun_out/$T
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest.txt"
tail -4 "$O/pytest.txt"
cd /tmp
for rep in 1 2; do
  for w in 4 8; do
    RNNOISE_AMD_GRU_W=$w python "$R/bench.py" --no-cpu-baseline --no-parity --repeats 9 --steps 20 > "$O/b.log" 2>&1
    grep '^{' "$O/b.log" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('gru_w=$w', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_standalone']['kernel_ms'])" | tee -a "$O/bench_ab.txt"
  done
done
for w in 4 8; do RNNOISE_AMD_GRU_W=$w python "$R/bench.py" --no-cpu-baseline --no-parity --repeats 5 --steps 20 --streams 16384 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('16384 gru_w=$w', d['value'], d['ms_per_step'])" | tee -a "$O/bench_ab.txt"; done
for w in 4 8; do RNNOISE_AMD_GRU_W=$w python "$R/bench.py" --no-cpu-baseline --no-parity --repeats 5 --steps 20 --model little --streams 32768 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('little 32768 gru_w=$w', d['value'], d['ms_per_step'])" | tee -a "$O/bench_ab.txt"; done
