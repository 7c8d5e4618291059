cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/final/pytest_gpu.txt 2>&1; tail -4 gpurun_out/final/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; tail -2 gpurun_out/final/smoke.txt
timeout 900 python tools/soak_long.py 417 1 2>&1 | grep -v amdgpu.ids > gpurun_out/final/soak_long.txt; cat gpurun_out/final/soak_long.txt
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1; tail -3 gpurun_out/collect.log
