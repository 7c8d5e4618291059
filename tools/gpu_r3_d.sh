#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3d
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.txt"
tail -5 "$O/pytest_gpu.txt"
timeout 300 rnnoise_amd/csrc/build/valu_issue cnd > "$O/valu_cnd.txt" 2>&1; cat "$O/valu_cnd.txt"
cd /tmp
python "$R/bench.py" > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_65536.json"; tail -3 "$O/b.log" | cut -c1-1500
python "$R/bench.py" --no-cpu-baseline --host-io --steps 8 --warmup 2 --repeats 9 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_hostio_65536.json"; tail -1 "$O/b.log" | cut -c1-400
python "$R/bench.py" --no-cpu-baseline --host-io --s16 --steps 8 --warmup 2 --repeats 9 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_hostio_s16_65536.json"; tail -1 "$O/b.log" | cut -c1-400
python "$R/bench.py" --no-cpu-baseline --s16 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_s16_65536.json"; tail -1 "$O/b.log" | cut -c1-400
python "$R/tools/serial_times.py" 1 16 64 1024 2>&1 | grep "N=" | tee "$O/serial_times_small.txt"
python "$R/tools/configs0.py" 2>&1 | grep configs | tee "$O/configs0.txt"
