# the 4,096-stream PMC pass of tools/collect_profiles.sh alone (eight-wave tile kernel), + checks of the 10,240-stream switch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES,SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES,SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT,TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum,FETCH_SIZE,WRITE_SIZE,GRBM_GUI_ACTIVE"
RNNOISE_AMD_TILE_WAVES=8 RNNOISE_AMD_PIPE=9 python "$R/tools/pmc_collect.py" "$O/pmc_4096" "$G" -- python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 8 --warmup 2 --repeats 2 > "$O/pmc_4096.csv" 2>&1
rm -rf "$O/pmc_4096"
cd $R
for n in 10240 12288 14336; do echo "default N=$n: $(python bench.py --no-cpu-baseline --streams $n --repeats 9 --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]/1e6,2), d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"], d[\"parity\"][\"bit_identical\"])")"; done
echo "tile N=14336: $(RNNOISE_AMD_NN_LAYERS_MIN=16384 python bench.py --no-cpu-baseline --streams 14336 --repeats 9 --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]/1e6,2), d[\"ms_per_step\"])")"
