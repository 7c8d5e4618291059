cd /tmp
p() { grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_min'], d['value_max'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for rep in 1 2; do echo "default"; python /root/repo/bench.py --no-cpu-baseline 2>&1 | p; done
echo "K1 LDS 10240 per wave"; RNNOISE_AMD_K1_LDS=10240 python /root/repo/bench.py --no-cpu-baseline 2>&1 | p
echo serial; RNNOISE_AMD_PIPE=9 python /root/repo/bench.py --no-cpu-baseline 2>&1 | p
echo 4096; python /root/repo/bench.py --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 2>&1 | p
echo little; python /root/repo/bench.py --no-cpu-baseline --model little --streams 32768 2>&1 | p
