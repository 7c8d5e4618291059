# after the last day's switch changes: the 4,096-stream PMC pass, the bench lines and the serial times once more on the final library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES,SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES,SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT,TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum,FETCH_SIZE,WRITE_SIZE,GRBM_GUI_ACTIVE"
RNNOISE_AMD_TILE_WAVES=8 RNNOISE_AMD_PIPE=9 python "$R/tools/pmc_collect.py" "$O/pmc_4096" "$G" -- python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 8 --warmup 2 --repeats 2 > "$O/pmc_4096.csv" 2>&1
rm -rf "$O/pmc_4096"
python "$R/tools/serial_times.py" 1 64 1024 4096 16384 65536 2>&1 | grep "N=" > "$O/serial_times.txt"
bash $R/tools/collect_bench_lines.sh
