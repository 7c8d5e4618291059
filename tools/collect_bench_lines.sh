#!/bin/bash
# Runs ON THE GPU BOX: only the bench lines of tools/collect_profiles.sh (second pass of a collection -- bench.py quotes the VALU counts, LDS
# cycles, traffic and clocks of profiles/pmc_by_streams.json and profiles/valu_mix.json, which the first pass has just refreshed).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof
mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
last() { grep '^{' "$1" | tail -1; }
python "$R/bench.py" > "$O/bench_65536.log" 2>&1;                                                        last "$O/bench_65536.log" > "$O/bench_65536.json"
python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 > "$O/b.log" 2>&1;          last "$O/b.log" > "$O/bench_4096.json"
python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 --nn vector > "$O/b.log" 2>&1; last "$O/b.log" > "$O/bench_4096_vector.json"
python "$R/bench.py" --no-cpu-baseline --model little --streams 32768 > "$O/b.log" 2>&1;                 last "$O/b.log" > "$O/bench_little_32768.json"
python "$R/bench.py" --no-cpu-baseline --streams 16384 --steps 40 --warmup 8 > "$O/b.log" 2>&1;          last "$O/b.log" > "$O/bench_16384.json"
python "$R/bench.py" --no-cpu-baseline --host-io --steps 12 --warmup 4 --repeats 7 > "$O/b.log" 2>&1;    last "$O/b.log" > "$O/bench_hostio_65536.json"
python "$R/bench.py" --no-cpu-baseline --host-io --s16 --steps 16 --warmup 4 --repeats 7 > "$O/b.log" 2>&1; last "$O/b.log" > "$O/bench_hostio_s16_65536.json"
python "$R/bench.py" --no-cpu-baseline --s16 > "$O/b.log" 2>&1;                                            last "$O/b.log" > "$O/bench_s16_65536.json"
python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 --frames-per-call 1 > "$O/b.log" 2>&1;  last "$O/b.log" > "$O/bench_4096_fpc1.json"
python "$R/bench.py" --no-cpu-baseline --streams 16384 --steps 40 --warmup 8 --frames-per-call 1 > "$O/b.log" 2>&1;  last "$O/b.log" > "$O/bench_16384_fpc1.json"
rm -f "$O/b.log"; ls -la "$O" | grep bench_
