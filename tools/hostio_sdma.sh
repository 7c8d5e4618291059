#!/bin/bash
# Runs ON THE GPU BOX: the host-fed path (pinned caller memory, PCIe inside the timed region) at 65,536 streams with its copy modes:
#   sdma  the two directions on two named copy engines at once (hsa_amd_memory_async_copy_on_engine underneath HIP; round 5)
#   one   uploads and downloads alternating on one HIP copy stream (round 3/4 default for int16)
#   hp    uploads on the high-pass stream, downloads on the copy stream (round 4 default for float: one of them becomes a blit kernel)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
last() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(f\"{d['value']/1e6:8.2f} M frames/s  {d['ms_per_step']:.3f} ms/step  (min {d['value_min']/1e6:.2f} max {d['value_max']/1e6:.2f})  parity {d.get('parity',{}).get('bit_identical')}\")"; }
for mode in sdma one hp; do
  echo "int16  copy mode $mode: $(RNNOISE_AMD_HOSTIO_COPY=$mode timeout 200 python $R/bench.py --no-cpu-baseline --host-io --s16 --steps 16 --warmup 4 --repeats 7 2>&1 | last)"
done
for mode in sdma one hp; do
  echo "float  copy mode $mode: $(RNNOISE_AMD_HOSTIO_COPY=$mode timeout 200 python $R/bench.py --no-cpu-baseline --host-io --steps 12 --warmup 4 --repeats 7 2>&1 | last)"
done
echo "HBM-resident input for comparison: $(timeout 200 python $R/bench.py --no-cpu-baseline --s16 --steps 40 --warmup 8 --repeats 7 2>&1 | last)"
