#!/bin/bash
# one K1 iteration on the GPU box: parity subset -> stand-alone kernel times -> bench line -> per-section profile
# usage: tools/gpu_k1_iter.sh TAG [full]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-k1it}
O=$R/gpurun_out/$T
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
if [ "${2:-}" = "full" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest.txt"
else
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py -x -q -k "taps or free_running or fuzz or golden or multi_stream or full_size or instrumented or register_fft or 16384" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest.txt"
fi
tail -5 "$O/pytest.txt"
cd /tmp
python "$R/tools/serial_times.py" 4096 16384 65536 2>&1 | grep "N=" | tee "$O/serial_times.txt"
python "$R/bench.py" --no-cpu-baseline --repeats 9 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_65536.json"
python -c "
import json; d=json.load(open('$O/bench_65536.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
bash "$R/tools/k1_prefix.sh" "$T/prefix" 65536 > /dev/null 2>&1
cat "$O/prefix/k1_prefix.txt"
