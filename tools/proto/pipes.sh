R=$GRAFT_REPO_ROOT; cd $R
b() { python bench.py --no-cpu-baseline --streams $1 --repeats 9 --steps 40 $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['bit_identical'])"; }
for n in 1024 2048 4096 8192 10240 16384 32768; do for p in 0 1 9; do echo "pipe=$p N=$n: $(RNNOISE_AMD_PIPE=$p b $n)"; done; done
