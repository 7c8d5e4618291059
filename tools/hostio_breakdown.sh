#!/bin/bash
# VERDICT r5 Next #7: what the host-fed path's step is made of.  A call pays one frame's upload before its first kernel and one frame's
# download after its last one, whatever its length: step(n) = a + b / n over calls of n frames separates the per-frame cost a (kernels of
# the host-fed schedule + whatever the copies beside them cost) from the per-call cost b; the kernels' own durations inside the call
# (HIP events) say how much of a is kernel time.  HBM-resident lines of the same schedule(s) beside them.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
line() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print(f\"{d['value']/1e6:8.2f} M frames/s  {d['ms_per_step']:.4f} ms/step  kernels inside: hp {k['highpass']:.3f} + K1 {k['analysis']:.3f} + K2 {k['network']:.3f} + K3 {k['synthesis']:.3f} = {k['analysis']+k['network']+k['synthesis']:.3f} on the main stream\")"; }
for n in 8 16 32 64; do
  echo "int16 host-fed, $n frames per call: $(timeout 200 python $R/bench.py --no-cpu-baseline --no-parity --host-io --s16 --steps $n --warmup 4 --repeats 7 2>&1 | line)"
done
for n in 8 16 32; do
  echo "float host-fed, $n frames per call: $(timeout 200 python $R/bench.py --no-cpu-baseline --no-parity --host-io --steps $n --warmup 4 --repeats 7 2>&1 | line)"
done
echo "int16 HBM-resident, schedule 1 (the host-fed path's: only the high-pass aside), 16 frames per call: $(RNNOISE_AMD_PIPE=1 timeout 200 python $R/bench.py --no-cpu-baseline --no-parity --s16 --steps 16 --warmup 4 --repeats 7 2>&1 | line)"
echo "int16 HBM-resident, default schedule (three streams), 16 frames per call: $(timeout 200 python $R/bench.py --no-cpu-baseline --no-parity --s16 --steps 16 --warmup 4 --repeats 7 2>&1 | line)"
python3 $R/tools/pcie_peak.py 2>&1 | grep -i pinned | head -4
