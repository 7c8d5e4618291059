cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/final/pytest_gpu.txt 2>&1; tail -4 gpurun_out/final/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/dense_fold_ab.sh final_fold 2>&1 | grep -E "passed|failed|variant=" | head -8
