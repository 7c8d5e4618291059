#!/bin/bash
# A/B of library BUILDS inside one gpurun call (boxes differ by several percent; only numbers from one call compare):
# every rnnoise_amd/librnnoise_amd_<tag>.so given as an argument is timed with tools/serial_times.py (stand-alone kernel times),
# round-robin for ROUNDS rounds, after the GPU parity tests have passed on the product library librnnoise_amd.so.
# usage: tools/ab_libs.sh <outdir under gpurun_out> <tag> [<tag> ...]      (RNNOISE_AMD_LIB selects the build: capi.py)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"; timeout 900 python -m pytest tests ${PYTEST_ARGS:--m gpu -x -q} 2>&1 | tail -3 | tee -a "$O/pytest.txt"
cd /tmp
for i in $(seq ${ROUNDS:-2}); do
  for v in "$@"; do
    RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_$v.so python "$R/tools/serial_times.py" ${STREAMS:-65536} 2>&1 | grep "N=" | sed "s/^/$v /" | tee -a "$O/serial.txt"
  done
done
for v in ${BENCH_TAGS:-}; do
  RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_$v.so python "$R/bench.py" --no-cpu-baseline | tail -1 > "$O/bench_$v.json"
  python - "$O/bench_$v.json" $v <<'PY' | tee -a "$O/bench.txt"
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity',{}).get('bit_identical'))
PY
done
echo done
