#!/bin/bash
# rocprofv3 evidence for bench.py's default workload (GPU box, repo root): kernel trace + stats, then separate PMC passes
# (each with --kernel-trace only).  usage: bash tools/profile_bench.sh gpurun_out/<tag> [extra bench.py flags]
OUT=${1:-gpurun_out/prof}; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
BENCH="python bench.py --steps ${STEPS:-100} --warmup ${WARMUP:-20} --blocks 3 --no-cpu-baseline --no-batched $*"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- $BENCH > "$OUT/bench_stats_run.json" 2> "$OUT/stats.log" || tail -3 "$OUT/stats.log"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_mfma" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc_mfma.log" || tail -3 "$OUT/pmc_mfma.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.log" || tail -3 "$OUT/pmc_fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc_write.log" || tail -3 "$OUT/pmc_write.log"
$BENCH > "$OUT/bench_default.json" 2>/dev/null
ls "$OUT" "$OUT/stats"
# keep only what make_profiles.py needs (the traces are large)
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
