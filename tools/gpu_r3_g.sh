#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g
rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O/trace" -- python "$R/bench.py" --no-cpu-baseline --no-parity --host-io --s16 --steps 16 --warmup 2 --repeats 2 > "$O/trace.log" 2>&1
tail -2 "$O/trace.log" | cut -c1-200
find "$O/trace" -name "*.csv" | xargs ls -la
