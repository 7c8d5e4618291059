# thresholds re-measured on the final kernels: analysis workgroup form (RNNOISE_AMD_K1_SPW), GRU layer kernel form (RNNOISE_AMD_GRU_VARIANT)
R=$GRAFT_REPO_ROOT; cd $R
b() { python bench.py --no-cpu-baseline --streams $1 --repeats 9 --steps 40 $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['bit_identical'])"; }
for r in 1 2; do
for n in 2048 3072 4096 5120; do
  echo "spw=1 N=$n: $(RNNOISE_AMD_K1_SPW=1 b $n)"; echo "spw=4 N=$n: $(RNNOISE_AMD_K1_SPW=4 b $n)"
done; done
for n in 3072 4096 5120; do
  echo "fpc1 spw=1 N=$n: $(RNNOISE_AMD_K1_SPW=1 b $n "--frames-per-call 1")"; echo "fpc1 spw=4 N=$n: $(RNNOISE_AMD_K1_SPW=4 b $n "--frames-per-call 1")"
done
for n in 12288 16384 20480 24576 32768; do
  echo "w4 N=$n: $(RNNOISE_AMD_GRU_VARIANT=w4 b $n)"; echo "w8 N=$n: $(RNNOISE_AMD_GRU_VARIANT=w8 b $n)"
done
