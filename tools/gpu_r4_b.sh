#!/bin/bash
# round 4, GPU call B: full GPU tests -> the drop-in call from C threads (combiner on / off) -> kernel times -> bench -> K1 sections
# usage: tools/gpu_r4_b.sh TAG [prefix]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r4b}
O=$R/gpurun_out/$T
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest.txt"
tail -5 "$O/pytest.txt"
gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$R/rnnoise_amd -lpthread
BLOB=oracle/_ref/default.blob
[ -f $BLOB ] || python -c "import lzma;open('/tmp/default.blob','wb').write(lzma.decompress(open('tests/golden/default.blob.xz','rb').read()))" && [ -f $BLOB ] || BLOB=/tmp/default.blob
for c in 1 0; do
  for t in 1 2 4 8 16 32 64; do
    RNNOISE_AMD_COMBINE=$c timeout 120 /tmp/configs0_mt $BLOB $t 3000 2>&1 | sed "s/^/combine=$c /" | tee -a "$O/configs0_cthreads.txt"
  done
done
for g in 0 5 30; do RNNOISE_AMD_COMBINE_GATHER_US=$g timeout 120 /tmp/configs0_mt $BLOB 16 3000 2>&1 | sed "s/^/gather_us=$g /" | tee -a "$O/configs0_cthreads.txt"; done
for s in 1 2 4; do RNNOISE_AMD_COMBINE_STREAMS=$s timeout 120 /tmp/configs0_mt $BLOB 16 3000 2>&1 | sed "s/^/streams=$s /" | tee -a "$O/configs0_cthreads.txt"; done
cd /tmp
python "$R/tools/serial_times.py" 1 64 16384 65536 2>&1 | grep "N=" | tee -a "$O/serial_times.txt"
python "$R/bench.py" --no-cpu-baseline --repeats 9 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_65536.json"
python -c "
import json
d=json.load(open('$O/bench_65536.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity',{}).get('bit_identical'))"
if [ "${2:-}" = "prefix" ]; then bash "$R/tools/k1_prefix.sh" "$T/prefix" 65536 > /dev/null 2>&1; cat "$O/prefix/k1_prefix.txt"; fi
