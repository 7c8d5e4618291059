cd /tmp
p() { grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_min'], d['value_max'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for l in 9504 13900 12800 18000; do echo "K1 LDS $l per wave"; RNNOISE_AMD_K1_LDS=$l python /root/repo/bench.py --no-cpu-baseline --repeats 9 2>&1 | p; done
