#!/bin/bash
# the narrow phases of rn_analysis_kernel in detail: stop points 6 / 7 (around phase 1 + coarse selection), 8 / 9 (around phase 2: fine chains beside
# the fine running energy), 10 (fine selection), 11 / 12 (around the paired candidate dots beside the yy_lookup sweep)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-k1_narrow}
mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
for k in 6 7 8 9 10 11 12; do
  rm -rf "$O/p"
  RNNOISE_AMD_K1_STOP=$k rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES \
      --output-format csv -d "$O/p" -- python "$R/tools/k1_prefix.py" 65536 4 > "$O/run_$k.log" 2>&1
  ms=$(grep -o "analysis_ms=[0-9.]*" "$O/run_$k.log" | cut -d= -f2)
  python - "$O/p" "$k" "$ms" <<'PY' | tee -a "$O/narrow.txt"
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("rn_analysis_kernel"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1) or 1
print(f"stop {sys.argv[2]:>3}: {sys.argv[3]} ms  valu/wave {m.get('SQ_INSTS_VALU',0)/w:.0f}  lds_cyc/wave {m.get('SQ_LDS_IDX_ACTIVE',0)/w:.0f}  conflicts {m.get('SQ_LDS_BANK_CONFLICT',0)/w:.0f}  wave_cycles/wave {4*m.get('SQ_WAVE_CYCLES',0)/w:.0f}  wait_any/wave {4*m.get('SQ_WAIT_ANY',0)/w:.0f}")
PY
done
rm -rf "$O/p"
