R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
b() { python bench.py --no-cpu-baseline --streams $1 --repeats 9 --steps 40 $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['bit_identical'])"; }
for n in 2048 2560 3072 4096; do echo "default N=$n: $(b $n)"; echo "spw=1   N=$n: $(RNNOISE_AMD_K1_SPW=1 b $n)"; done
for n in 2560 4096; do echo "fpc1 default N=$n: $(b $n "--frames-per-call 1")"; echo "fpc1 spw=1   N=$n: $(RNNOISE_AMD_K1_SPW=1 b $n "--frames-per-call 1")"; done
