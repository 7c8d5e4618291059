#!/bin/bash
# A/B of the analysis-kernel dispatch choices (streams per workgroup, 80-VGPR build) across batch sizes, pipelined bench
cd /tmp
R=/root/repo
for n in 2048 4096 8192 16384; do
  for spw in 1 4; do for lean in 0 1; do
    v=$(RNNOISE_AMD_K1_SPW=$spw RNNOISE_AMD_K1_LEAN=$lean python $R/bench.py --no-cpu-baseline --streams $n --steps 40 --warmup 8 --repeats 9 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])")
    echo "n=$n spw=$spw lean=$lean : $v"
  done; done
done
