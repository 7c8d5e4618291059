#!/bin/bash
# PMC diagnosis passes at 65536 streams (serial schedule, so that kernels do not overlap): where do the cycles of a kernel go
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_k1}
mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
G="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY,SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA,SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"
RNNOISE_AMD_PIPE=9 python "$R/tools/pmc_collect.py" "$O/p" "$G" -- python "$R/bench.py" --no-cpu-baseline --streams ${2:-65536} --steps 4 --warmup 1 --repeats 2 > "$O/pmc.csv" 2>&1
rm -rf "$O/p"
python - "$O/pmc.csv" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith('#'))]
for r in rows:
    g=lambda k: float(r.get(k) or 0)
    w=g('SQ_WAVES') or 1
    print(f"{r['kernel']:<26} waves {w:8.0f} | per wave: VALU {g('SQ_INSTS_VALU')/w:7.0f} SALU {g('SQ_INSTS_SALU')/w:6.0f} LDS {g('SQ_INSTS_LDS')/w:6.0f} VMEM {(g('SQ_INSTS_VMEM_RD')+g('SQ_INSTS_VMEM_WR'))/w:5.0f} "
          f"| wave cycles {4*g('SQ_WAVE_CYCLES')/w:9.0f} valu-active {4*g('SQ_ACTIVE_INST_VALU')/w:8.0f} wait_any {100*g('SQ_WAIT_ANY')/(g('SQ_WAVE_CYCLES') or 1):5.1f}% wait_inst {100*g('SQ_WAIT_INST_ANY')/(g('SQ_WAVE_CYCLES') or 1):5.1f}% "
          f"| LDS active cyc/CU {g('SQ_LDS_IDX_ACTIVE')/256:10.0f} bank-conflict cyc/CU {g('SQ_LDS_BANK_CONFLICT')/256:10.0f} ({100*g('SQ_LDS_BANK_CONFLICT')/(g('SQ_LDS_IDX_ACTIVE') or 1):4.1f}%) per wave {g('SQ_LDS_IDX_ACTIVE')/w:7.0f} "
          f"| MFMA insts/wave {g('SQ_INSTS_VALU_MFMA_I8')/w:6.0f} mfma-busy/busy {100*g('SQ_VALU_MFMA_BUSY_CYCLES')/(g('SQ_BUSY_CYCLES') or 1):5.1f}% vmem-active {4*g('SQ_ACTIVE_INST_VMEM')/w:8.0f}")
PY
