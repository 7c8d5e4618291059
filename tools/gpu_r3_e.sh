#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3e
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
timeout 300 rnnoise_amd/csrc/build/valu_issue cnd > "$O/valu_cnd.txt" 2>&1; cat "$O/valu_cnd.txt" | grep -v "^#"
timeout 600 python -m pytest tests -m gpu -x -q -k "host_fed or s16 or cli" 2>&1 | tail -3
cd /tmp
python "$R/tools/pcie_peak.py" 2>&1 | tee "$O/pcie_peak.txt"
python "$R/tools/pcie_peak.py" 504 2>&1 | tee -a "$O/pcie_peak.txt"
python "$R/bench.py" --no-cpu-baseline --host-io --steps 16 --warmup 4 --repeats 7 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_hostio_65536.json"; tail -1 "$O/b.log" | cut -c1-300
python "$R/bench.py" --no-cpu-baseline --host-io --s16 --steps 16 --warmup 4 --repeats 7 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_hostio_s16_65536.json"; tail -1 "$O/b.log" | cut -c1-300
python "$R/bench.py" --no-cpu-baseline --host-io --s16 --steps 32 --warmup 4 --repeats 5 > "$O/b.log" 2>&1; grep '^{' "$O/b.log" | tail -1 > "$O/bench_hostio_s16_65536_k32.json"; tail -1 "$O/b.log" | cut -c1-300
