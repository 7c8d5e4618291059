#!/bin/bash
# K1 by section (on the GPU box): rn_analysis_kernel leaves at stop point k = 1..16 (0 = whole kernel); per stop point one
# rocprofv3 --pmc pass.  Differences between consecutive rows = that section's LDS cycles / conflicts / instructions / time.
# usage: tools/k1_prefix.sh [outdir under gpurun_out] [streams]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-k1_prefix}
N=${2:-65536}
mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
NAMES=(whole "window X" "FFT X" "store X + Ex" "downsample+FIR" "y4/Z + coarse pass 1" "coarse pass 2" "narrow 1 + coarse select" "shifted copy" "narrow 2+3" "fine select" "doubling prep" "doubling dots" "decide" "window P + loads" "FFT P" "store P + Ep + Exp" )
echo "stop,section,ms,waves,valu_per_wave,salu_per_wave,lds_inst_per_wave,lds_cyc_per_wave,conflict_cyc_per_wave,wave_cycles,wait_lds_pct" > "$O/k1_prefix.csv"
for k in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 0; do
  rm -rf "$O/p"
  RNNOISE_AMD_K1_STOP=$k rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_LDS \
      --output-format csv -d "$O/p" -- python "$R/tools/k1_prefix.py" $N 4 > "$O/run_$k.log" 2>&1
  ms=$(grep -o "analysis_ms=[0-9.]*" "$O/run_$k.log" | cut -d= -f2)
  python - "$O/p" "$k" "${NAMES[$k]}" "$ms" >> "$O/k1_prefix.csv" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("rn_analysis_kernel"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in acc.items()}   # skip the two warm-up launches
w = m.get("SQ_WAVES", 1) or 1
print(f"{sys.argv[2]},{sys.argv[3]},{sys.argv[4]},{w:.0f},{m.get('SQ_INSTS_VALU',0)/w:.0f},{m.get('SQ_INSTS_SALU',0)/w:.0f},{m.get('SQ_INSTS_LDS',0)/w:.0f},"
      f"{m.get('SQ_LDS_IDX_ACTIVE',0)/w:.0f},{m.get('SQ_LDS_BANK_CONFLICT',0)/w:.0f},{4*m.get('SQ_WAVE_CYCLES',0)/w:.0f},{100*m.get('SQ_WAIT_INST_LDS',0)/(m.get('SQ_WAVE_CYCLES',1) or 1):.1f}")
PY
done
rm -rf "$O/p"
python - "$O/k1_prefix.csv" > "$O/k1_prefix.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
whole = rows[-1]
rows = rows[:-1] + [dict(whole, stop="17", section="features (rest)")]
print(f"# rn_analysis_kernel by section: the kernel is run 17 times, its workgroups leaving at stop point k (instrumented build, K1_STOP);")
print(f"# a row = stop k minus stop k-1.  ms = HIP-event duration; cycles per wave from SQ_* PMC counters (one pass per stop point).")
print(f"{'section':<28}{'d ms':>8}{'cum ms':>8}{'VALU':>7}{'SALU':>7}{'LDSinst':>8}{'LDScyc':>8}{'conflict':>9}{'cyc/LDSinst':>12}")
prev = dict(ms=0, valu_per_wave=0, salu_per_wave=0, lds_inst_per_wave=0, lds_cyc_per_wave=0, conflict_cyc_per_wave=0)
for r in rows:
    g = lambda k: float(r[k] or 0)
    d = {k: g(k) - float(prev[k]) for k in prev}
    li = d["lds_inst_per_wave"]
    print(f"{r['section']:<28}{d['ms']:>8.3f}{g('ms'):>8.3f}{d['valu_per_wave']:>7.0f}{d['salu_per_wave']:>7.0f}{li:>8.0f}{d['lds_cyc_per_wave']:>8.0f}{d['conflict_cyc_per_wave']:>9.0f}"
          f"{(d['lds_cyc_per_wave'] / li if li > 0 else 0):>12.1f}")
    prev = {k: g(k) for k in prev}
print(f"{'whole kernel':<28}{'':>8}{float(whole['ms']):>8.3f}{float(whole['valu_per_wave']):>7.0f}{float(whole['salu_per_wave']):>7.0f}{float(whole['lds_inst_per_wave']):>8.0f}"
      f"{float(whole['lds_cyc_per_wave']):>8.0f}{float(whole['conflict_cyc_per_wave']):>9.0f}")
PY
cat "$O/k1_prefix.txt"
