#!/bin/bash
# Runs ON THE GPU BOX: does a copy of the host-fed path run as a blit KERNEL on the CUs?  rocprofv3 --kernel-trace --stats of
# `bench.py --host-io --s16` (65,536 streams) per copy mode; listed: the runtime's copy kernels (__amd_rocclr_*) and the analysis kernel beside them.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp
for mode in sdma hp one; do
  rm -rf /tmp/tr_$mode
  RNNOISE_AMD_HOSTIO_COPY=$mode rocprofv3 --kernel-trace --stats -d /tmp/tr_$mode -- python $R/bench.py --no-cpu-baseline --no-parity --host-io --s16 --steps 16 --warmup 4 --repeats 5 > /tmp/tr_$mode.log 2>&1
  echo "---- copy mode $mode: $(grep '^{' /tmp/tr_$mode.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(f\"{d['value']/1e6:.2f} M frames/s, {d['ms_per_step']:.3f} ms/step\")")"
  python $R/tools/prof_summary.py "$(ls /tmp/tr_$mode/*/*_results.db | head -1)" "mode $mode" | grep -E "^kernel|rocclr|rn_analysis_kernel|rn_release_store|rn_hp" | cut -c1-130
done
