#!/bin/bash
# A/B of library builds in one call: prev = HEAD before the change, pair = paired doubling dots only, new = working tree
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4m}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a "$O/pytest.txt"
if ! grep -q " passed" "$O/pytest.txt" || grep -q "failed" "$O/pytest.txt"; then
  RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_pair.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | sed "s/^/pair-lib /" | tee -a "$O/pytest.txt"
fi
cd /tmp
for i in 1 2; do
  for v in prev pair new; do
    L=$R/rnnoise_amd/librnnoise_amd_$v.so; [ $v = new ] && L=$R/rnnoise_amd/librnnoise_amd.so
    RNNOISE_AMD_LIB=$L python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/$v /" | tee -a "$O/serial.txt"
  done
  RNNOISE_AMD_K1_EXPERIMENT=2097152 python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/new+deep /" | tee -a "$O/serial.txt"
  RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_pair.so RNNOISE_AMD_K1_EXPERIMENT=2097152 python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/pair+deep /" | tee -a "$O/serial.txt"
done
for v in prev new; do
  L=$R/rnnoise_amd/librnnoise_amd_$v.so; [ $v = new ] && L=$R/rnnoise_amd/librnnoise_amd.so
  RNNOISE_AMD_LIB=$L python "$R/bench.py" --no-cpu-baseline 2>/dev/null | tail -1 > "$O/bench_$v.json" || RNNOISE_AMD_LIB=$L python "$R/bench.py" | tail -1 > "$O/bench_$v.json"
  python - "$O/bench_$v.json" $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity',{}).get('bit_identical'))
PY
done | tee -a "$O/bench.txt"
echo done
