#!/bin/bash
# host-fed path experiments (on the GPU box) -- both copy directions at once under different SDMA settings of the runtime
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r4f}
O=$R/gpurun_out/$T
export RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_instr.so  # (the A/B switches below exist in the instrumented library only)
mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
run() {  # tag, env..., -- bench args
  local tag=$1; shift
  ( env "$@" python "$R/bench.py" --no-cpu-baseline --no-parity --host-io --steps 16 --warmup 4 --repeats 5 $EXTRA > "$O/$tag.log" 2>&1
    grep '^{' "$O/$tag.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '$EXTRA', d['value'], d['ms_per_step'])" ) 2>&1 | tee -a "$O/hostio.txt"
}
for EXTRA in "--s16" ""; do
  run one_stream X=1
  run two_streams RNNOISE_AMD_HOSTIO_COPY=hp
  run two_streams_nogang RNNOISE_AMD_HOSTIO_COPY=hp HSA_ENABLE_SDMA_GANG=0
  run one_stream_nogang HSA_ENABLE_SDMA_GANG=0
  run two_streams_sched0 RNNOISE_AMD_HOSTIO_COPY=hp RNNOISE_AMD_HOSTIO_SCHEDULE=0 HSA_ENABLE_SDMA_GANG=0
done
python "$R/tools/pcie_peak.py" 2>&1 | grep pinned | tee "$O/pcie_peak.txt"
HSA_ENABLE_SDMA_GANG=0 python "$R/tools/pcie_peak.py" 2>&1 | grep pinned | sed "s/^/nogang /" | tee -a "$O/pcie_peak.txt"
