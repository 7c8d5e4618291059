#!/bin/bash
# Runs ON THE GPU BOX (gpurun): tests + the bench lines of the three single-GPU configs + rocprof stats + section taps.
#   tools/gpu_run1.sh <tag>      -> gpurun_out/<tag>/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-run}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd "$R"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.txt"
  tail -5 "$O/pytest_gpu.txt"
fi
cd /tmp
last() { grep '^{' "$1" | tail -1; }
timeout 600 python "$R/bench.py" > "$O/bench_65536.log" 2>&1;  last "$O/bench_65536.log" > "$O/bench_65536.json"
timeout 600 python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 > "$O/b.log" 2>&1; last "$O/b.log" > "$O/bench_4096.json"
timeout 600 python "$R/bench.py" --no-cpu-baseline --model little --streams 32768 > "$O/b.log" 2>&1; last "$O/b.log" > "$O/bench_little_32768.json"
timeout 600 python "$R/tools/serial_times.py" 4096 65536 2>&1 | grep "N=" > "$O/serial_times.txt"
timeout 600 python "$R/tools/k1_cycles.py" 65536 --nn > "$O/cycles_65536.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace" -- python "$R/bench.py" --no-cpu-baseline --repeats 5 > "$O/trace.log" 2>&1
python "$R/tools/prof_summary.py" "$(ls "$O"/trace/*/*_results.db | head -1)" \
  "python bench.py --no-cpu-baseline --repeats 5  [configs[2]: 65536 streams, MFMA network path, 3-stream pipeline]" > "$O/kernel_stats.txt"
rm -rf "$O/trace" "$O/b.log"
cat "$O/bench_65536.json" "$O/bench_4096.json" "$O/bench_little_32768.json" "$O/serial_times.txt" "$O/kernel_stats.txt"
