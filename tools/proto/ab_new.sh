export PYTEST_ARGS="tests/test_gpu_parity.py tests/test_dropin_gpu.py tests/test_train_features.py -m gpu -x -q"
ROUNDS=3 STREAMS="1 64 1024 4096 16384 65536" bash tools/ab_libs.sh new base new
cd $GRAFT_REPO_ROOT
for v in base new base new; do
  RNNOISE_AMD_LIB=$PWD/rnnoise_amd/librnnoise_amd_$v.so python bench.py --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 4096', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['bit_identical'])" | tee -a gpurun_out/new/bench.txt
  RNNOISE_AMD_LIB=$PWD/rnnoise_amd/librnnoise_amd_$v.so python bench.py --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 --frames-per-call 1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 4096 fpc1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a gpurun_out/new/bench.txt
  RNNOISE_AMD_LIB=$PWD/rnnoise_amd/librnnoise_amd_$v.so python bench.py --no-cpu-baseline | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 65536', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['bit_identical'])" | tee -a gpurun_out/new/bench.txt
  RNNOISE_AMD_LIB=$PWD/rnnoise_amd/librnnoise_amd_$v.so python tools/configs0.py 2>&1 | grep "pooled         1 thread" | sed "s/^/$v /" | tee -a gpurun_out/new/bench.txt
done
