#!/bin/bash
# VERDICT r5 Next #5a, bounded: what would rn_analysis_kernel take if the LDS bank conflicts of remove_doubling's candidate dots (495 of that
# pass's 985 LDS cycles per wave, profiles/r5_k1_sections.txt) were gone?  The instrumented library can run the pass with offsets 2 l -- one bank
# pair per lane, no conflict possible, WRONG RESULTS -- which is the floor of anything a different base layout (a skewed second copy, ...) could
# reach, since the real offsets are data.  Stand-alone kernel times at 65,536 streams, alternating, one gpurun call.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export RNNOISE_AMD_LIB=$R/rnnoise_amd/librnnoise_amd_instr.so
for i in 1 2 3; do
  echo "as shipped:                        $(timeout 120 python $R/tools/serial_times.py 65536 2>&1 | tail -1)"
  echo "candidate dots at conflict-free offsets: $(RNNOISE_AMD_K1_EXPERIMENT=8192 timeout 120 python $R/tools/serial_times.py 65536 2>&1 | tail -1)"
done
