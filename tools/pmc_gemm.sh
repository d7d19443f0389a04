#!/bin/bash
# rocprofv3 PMC passes of one GEMM configuration alone (tools/gemm_pipe_ab.py, one shape set).  Counter passes carry --kernel-trace only.
# usage (GPU box, repo root): bash tools/pmc_gemm.sh <out-dir under gpurun_out> <cfg> <shape set> [kernel regex]
OUT=${1:-gpurun_out/pmc_gemm}; CFG=${2:-30}; SH=${3:-KS}; RE=${4:-gemm_pipe}
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
for ctrs in \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
  "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "$RE" -d "$OUT/p$i" -o a --output-format csv -- \
      python tools/gemm_pipe_ab.py --epi bf16 --cfgs $CFG --shapes $SH --rounds 1 --iters 3 > "$OUT/p$i.log" 2>&1 || tail -3 "$OUT/p$i.log"
done
python - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(out + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]].add(r["Dispatch_Id"])
for key in sorted(agg):
    print(key)
    for c in sorted(agg[key]):
        print("   %-36s %16.0f per launch" % (c, agg[key][c] / max(1, len(n[key][c]))))
PY
