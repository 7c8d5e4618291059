# rocprofv3 kernel-trace stats of the configs[1] batch (4,096 streams) on the last library: pipelined calls and one frame per call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof4096; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d "$O/trace" -- python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 --repeats 5 > "$O/trace.log" 2>&1
python "$R/tools/prof_summary.py" "$(ls "$O"/trace/*/*_results.db | head -1)" "python bench.py --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 --repeats 5  [configs[1]'s batch: 4096 streams, MFMA network path, 3-stream pipeline + one stand-alone pass]" > "$O/kernel_stats_4096.txt"
rocprofv3 --kernel-trace --stats -d "$O/trace1" -- python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 --repeats 5 --frames-per-call 1 > "$O/trace1.log" 2>&1
python "$R/tools/prof_summary.py" "$(ls "$O"/trace1/*/*_results.db | head -1)" "python bench.py --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 --repeats 5 --frames-per-call 1  [one frame per call]" > "$O/kernel_stats_4096_fpc1.txt"
rm -rf "$O/trace" "$O/trace1"; grep "^rn_\|^kernel" "$O/kernel_stats_4096.txt" "$O/kernel_stats_4096_fpc1.txt" | cut -c1-150
