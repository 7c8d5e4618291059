#!/bin/bash
# A/B of environment switches inside one gpurun call: every argument after the output directory is "label|VAR=value VAR=value ..."
# (empty list = the defaults); each is timed with bench.py (--repeats 9, no CPU leg), round-robin for ROUNDS rounds.
# LIB=instr (default) runs the instrumented library (where the lab switches live), LIB=<tag> rnnoise_amd/librnnoise_amd_<tag>.so,
# LIB=product the product.  BENCH_ARGS adds bench.py arguments (e.g. --streams 16384).
# usage: tools/env_ab.sh <outdir under gpurun_out> "default|" "wpb4|RNNOISE_AMD_HP_WPB=4" ...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
case "${LIB:-instr}" in
  product) ;;
  *) export RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_${LIB:-instr}.so ;;
esac
last() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(f\"{d['value']/1e6:8.2f} M frames/s  {d['ms_per_step']:.4f} ms/step  (min {d['value_min']/1e6:.2f} max {d['value_max']/1e6:.2f})  parity {d.get('parity',{}).get('bit_identical')}  inside the pipeline: {d['roofline']['kernel_ms']}\")"; }
for i in $(seq ${ROUNDS:-3}); do
  for spec in "$@"; do
    label=${spec%%|*}; envs=${spec#*|}
    printf "%-28s %s\n" "$label" "$(env $envs timeout 300 python bench.py --no-cpu-baseline --repeats 9 ${BENCH_ARGS:-} 2>&1 | last)" | tee -a "$O/env_ab.txt"
  done
done
echo done
