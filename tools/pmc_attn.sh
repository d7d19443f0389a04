#!/bin/bash
# rocprofv3 PMC passes of the fused attention kernel alone (tools/attn_bench.py, one shape).  Counter passes carry --kernel-trace only.
# usage (GPU box, repo root): bash tools/pmc_attn.sh <out-dir under gpurun_out> [B H N cfg]
OUT=${1:-gpurun_out/pmc_attn}
B=${2:-32}; H=${3:-16}; N=${4:-681}; CFG=${5:--1}
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
for ctrs in \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "attn" -d "$OUT/p$i" -o a --output-format csv -- \
      python tools/attn_bench.py $B $H $N --cfgs $CFG > "$OUT/p$i.log" 2>&1 || tail -5 "$OUT/p$i.log"
done
python tools/pmc_report.py "$OUT" attn | tee "$OUT/summary.txt"
grep "^attention" "$OUT/p1.log" | tee -a "$OUT/summary.txt"
