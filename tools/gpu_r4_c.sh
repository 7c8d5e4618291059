#!/bin/bash
# round 4, GPU call C: drop-in tests + C threads (combiner) + K1 A/B (xrow, spread) + GRU prefetch depth A/B, all in one box
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r4c}
O=$R/gpurun_out/$T
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest.txt"
tail -5 "$O/pytest.txt"
gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$R/rnnoise_amd -lpthread
BLOB=oracle/_ref/default.blob
for t in 1 2 4 8 16 32 64; do timeout 120 /tmp/configs0_mt $BLOB $t 3000 2>&1 | sed "s/^/combine=1 /" | tee -a "$O/configs0_cthreads.txt"; done
for t in 4 16; do RNNOISE_AMD_COMBINE_POLL=0 timeout 120 /tmp/configs0_mt $BLOB $t 3000 2>&1 | sed "s/^/poll=0 /" | tee -a "$O/configs0_cthreads.txt"; done
for g in 0 40; do RNNOISE_AMD_COMBINE_GATHER_US=$g timeout 120 /tmp/configs0_mt $BLOB 16 3000 2>&1 | sed "s/^/gather_us=$g /" | tee -a "$O/configs0_cthreads.txt"; done
for s in 1 2 4; do RNNOISE_AMD_COMBINE_STREAMS=$s timeout 120 /tmp/configs0_mt $BLOB 16 3000 2>&1 | sed "s/^/streams=$s /" | tee -a "$O/configs0_cthreads.txt"; done
cd /tmp
for rep in 1 2; do
  python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/default /" | tee -a "$O/serial_times.txt"
  RNNOISE_AMD_K1_XROW=0 python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/xrow=0 /" | tee -a "$O/serial_times.txt"
  RNNOISE_AMD_K1_SPREAD=0 python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/spread=0 /" | tee -a "$O/serial_times.txt"
  RNNOISE_AMD_GRU_AD=3 python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/gru_ad=3 /" | tee -a "$O/serial_times.txt"
  RNNOISE_AMD_GRU_AD=4 python "$R/tools/serial_times.py" 65536 2>&1 | grep "N=" | sed "s/^/gru_ad=4 /" | tee -a "$O/serial_times.txt"
done
python "$R/bench.py" --no-cpu-baseline --repeats 9 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_65536.json"
python -c "
import json
d=json.load(open('$O/bench_65536.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity',{}).get('bit_identical'))"
if [ "${2:-}" = "prefix" ]; then bash "$R/tools/k1_prefix.sh" "$T/prefix" 65536 > /dev/null 2>&1; cat "$O/prefix/k1_prefix.txt"; fi
