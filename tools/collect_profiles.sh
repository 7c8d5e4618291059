#!/bin/bash
# Runs ON THE GPU BOX (gpurun): produces every artefact kept under profiles/ into gpurun_out/prof/.
#   tools/collect_profiles.sh   -> bench lines (configs[1..3], host-fed float / int16, one frame per call), rocprofv3 kernel stats,
#                                  PMC passes, section taps, the VALU issue table and the PCIe yardstick
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
last() { grep '^{' "$1" | tail -1; }
python "$R/bench.py" > "$O/bench_65536.log" 2>&1;                                                        last "$O/bench_65536.log" > "$O/bench_65536.json"
python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 > "$O/b.log" 2>&1;          last "$O/b.log" > "$O/bench_4096.json"
python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 --nn vector > "$O/b.log" 2>&1; last "$O/b.log" > "$O/bench_4096_vector.json"
python "$R/bench.py" --no-cpu-baseline --model little --streams 32768 > "$O/b.log" 2>&1;                 last "$O/b.log" > "$O/bench_little_32768.json"
python "$R/bench.py" --no-cpu-baseline --streams 16384 --steps 40 --warmup 8 > "$O/b.log" 2>&1;          last "$O/b.log" > "$O/bench_16384.json"
python "$R/bench.py" --no-cpu-baseline --host-io --steps 12 --warmup 4 --repeats 7 > "$O/b.log" 2>&1;    last "$O/b.log" > "$O/bench_hostio_65536.json"
python "$R/bench.py" --no-cpu-baseline --host-io --s16 --steps 16 --warmup 4 --repeats 7 > "$O/b.log" 2>&1; last "$O/b.log" > "$O/bench_hostio_s16_65536.json"
python "$R/bench.py" --no-cpu-baseline --s16 > "$O/b.log" 2>&1;                                            last "$O/b.log" > "$O/bench_s16_65536.json"
python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 50 --warmup 10 --frames-per-call 1 > "$O/b.log" 2>&1;  last "$O/b.log" > "$O/bench_4096_fpc1.json"
python "$R/bench.py" --no-cpu-baseline --streams 16384 --steps 40 --warmup 8 --frames-per-call 1 > "$O/b.log" 2>&1;  last "$O/b.log" > "$O/bench_16384_fpc1.json"
python "$R/tools/pcie_peak.py" 2>&1 | grep pinned > "$O/pcie_peak.txt"
python "$R/tools/serial_times.py" 1 64 1024 4096 16384 65536 2>&1 | grep "N=" > "$O/serial_times.txt"
# (the VALU issue table does not depend on the library: COLLECT_VALU_ISSUE=1 re-measures it, otherwise the last round's stays)
[ "${COLLECT_VALU_ISSUE:-0}" = 1 ] && timeout 600 "$R/rnnoise_amd/csrc/build/valu_issue" > "$O/valu_issue.txt" 2>&1
RNNOISE_AMD_NN_LAYERS_MIN=100000000 python "$R/tools/ab_layers.py" 65536 2>&1 | grep -E "^N=|^n=" > "$O/network_schedules_65536.txt"
python "$R/tools/k1_cycles.py" 65536 --nn --layers 2>&1 | grep -v amdgpu.ids > "$O/section_taps_65536.txt"
python "$R/tools/configs0.py" 2>&1 | grep configs > "$O/configs0.txt"
# the drop-in call from plain C threads (the combiner of dropin.cpp): throughput, CPU time per frame, more states than threads, pool sizes
bash "$R/tools/configs0_scope.sh" 2>&1 | grep -v amdgpu.ids > "$O/configs0_cthreads.txt"
# the host-fed path's copy modes (round 5) and what its step is made of (round 6)
bash "$R/tools/hostio_sdma.sh" 2>&1 | grep -v amdgpu.ids > "$O/hostio_sdma.txt"
bash "$R/tools/hostio_breakdown.sh" 2>&1 | grep -v amdgpu.ids > "$O/hostio_breakdown.txt"
# round 6: what rn_analysis_kernel would take without the bank conflicts of its candidate dots (instrumented library, timing only)
bash "$R/tools/k1_dots_conflicts.sh" 2>&1 | grep -v amdgpu.ids > "$O/k1_dots_conflicts.txt"
# round 5's investigations (the layer kernel's lab forms, what the frame pipeline hides): COLLECT_R5=1 repeats them on the instrumented library
if [ "${COLLECT_R5:-0}" = 1 ]; then
  python "$R/tools/gru_variants.py" w4 w8 p v3 2>&1 | grep -v amdgpu.ids > "$O/gru_variants.txt"
  python "$R/tools/overlap_table.py" 2>&1 | grep -v amdgpu.ids > "$O/overlap.txt"
fi
python "$R/tools/fft_bench.py" 2>&1 | grep -v amdgpu.ids > "$O/fft_bench.txt"
rocprofv3 --kernel-trace --stats -d "$O/trace" -- python "$R/bench.py" --no-cpu-baseline --repeats 5 > "$O/trace.log" 2>&1
python "$R/tools/prof_summary.py" "$(ls "$O"/trace/*/*_results.db | head -1)" \
  "python bench.py --no-cpu-baseline --repeats 5  [configs[2]: 65536 streams, MFMA network path, 3-stream pipeline + one stand-alone pass]" > "$O/kernel_stats.txt"
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES,SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES,SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT,TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum,FETCH_SIZE,WRITE_SIZE,GRBM_GUI_ACTIVE"
# PMC passes on the one-stream schedule (RNNOISE_AMD_PIPE=9): a kernel's counters are then its own, not a neighbour's
RNNOISE_AMD_PIPE=9 python "$R/tools/pmc_collect.py" "$O/pmc_65536" "$G" -- python "$R/bench.py" --no-cpu-baseline --steps 4 --warmup 1 --repeats 2 > "$O/pmc_65536.csv" 2>&1
# (the one-stream schedule counts as "nothing beside the network": it would take the sixteen-wave tile kernel of one-frame calls; the record
#  wanted is that of the eight-wave kernel pipelined calls run)
RNNOISE_AMD_TILE_WAVES=8 RNNOISE_AMD_PIPE=9 python "$R/tools/pmc_collect.py" "$O/pmc_4096" "$G" -- python "$R/bench.py" --no-cpu-baseline --streams 4096 --steps 8 --warmup 2 --repeats 2 > "$O/pmc_4096.csv" 2>&1
RNNOISE_AMD_PIPE=9 python "$R/tools/pmc_collect.py" "$O/pmc_little_32768" "$G" -- python "$R/bench.py" --no-cpu-baseline --model little --streams 32768 --steps 4 --warmup 1 --repeats 2 > "$O/pmc_little_32768.csv" 2>&1
bash "$R/tools/k1_prefix.sh" prof/k1_prefix 65536 > /dev/null 2>&1
cp "$O/k1_prefix/k1_prefix.txt" "$O/k1_sections.txt"; cp "$O/k1_prefix/k1_prefix.csv" "$O/k1_sections.csv"
# the narrow phases once more with the wave-cycle and wait counters (tools/k1_narrow.sh: stop points 6 .. 12)
bash "$R/tools/k1_narrow.sh" prof/k1_narrow > /dev/null 2>&1
{ echo "# rn_analysis_kernel narrow phases (tools/k1_narrow.sh, 65,536 streams): the kernel with its workgroups leaving at stop point k."
  echo "# 6 -> 7: barrier | coarse running energy on one wave | barrier + coarse selection;  8 -> 9: barrier | fine chains || start energy + fine Syy | barrier;"
  echo "# 9 -> 10: fine selection;  11 -> 12: candidate dots on two waves || yy_lookup | barrier   (11 = behind the barrier that posts T0)"
  cat "$O/k1_narrow/narrow.txt"; } > "$O/k1_narrow.txt"
rm -rf "$O/k1_narrow"
rm -rf "$O"/pmc_65536 "$O"/pmc_4096 "$O"/pmc_little_32768 "$O"/trace "$O"/b.log "$O"/k1_prefix
ls -la "$O"
