#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6g}
mkdir -p "$O"; export TMPDIR=/tmp
cd "$R"
bash tools/fused_k3k1.sh ${1:-r6g} 2>&1 | tail -14
( bash tools/hostio_breakdown.sh ) > "$O/hostio_breakdown.txt" 2>&1; cat "$O/hostio_breakdown.txt"
bash tools/dense_fold_ab.sh ${1:-r6g}_fold 2>&1 | tail -8
