#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f
mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
BENCH_TRACE=1 python "$R/bench.py" --no-cpu-baseline --host-io --s16 --steps 16 --warmup 4 --repeats 7 2>&1 | cut -c1-200 | tail -8
