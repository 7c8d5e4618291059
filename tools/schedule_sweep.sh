#!/bin/bash
# the frame pipeline's switches against each other inside one gpurun call (bench.py, 65,536 streams, pipelined multi-frame calls):
#   default (three streams: K0 two frames ahead, K1 of the next frame beside K2 + K3) | RNNOISE_AMD_PIPE=1 (only K0 aside) | =9 (one stream)
#   | RNNOISE_AMD_K1_PRIO=0 | RNNOISE_AMD_HP_EARLY=1
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-sched}
export RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_instr.so  # (the A/B switches below exist in the instrumented library only)
mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
run() { local tag=$1; shift
  env "$@" python "$R/bench.py" --no-cpu-baseline --no-parity --repeats 11 ${EXTRA:-} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '${EXTRA:-}', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a "$O/sweep.txt"; }
for i in 1 2; do
  run default X=1
  run pipe1 RNNOISE_AMD_PIPE=1
  run pipe9 RNNOISE_AMD_PIPE=9
  run k1prio0 RNNOISE_AMD_K1_PRIO=0
  run hp_early RNNOISE_AMD_HP_EARLY=1
done
EXTRA="--streams 16384 --steps 40 --warmup 8"
for i in 1 2; do run default X=1; run pipe1 RNNOISE_AMD_PIPE=1; run pipe9 RNNOISE_AMD_PIPE=9; done
echo done
