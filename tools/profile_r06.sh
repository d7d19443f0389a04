#!/bin/bash
# Round-6 evidence run (GPU box, repo root): the LayerNorm-free one-sequence frame against the LayerNorm-kernel schedule on ONE box (interleaved frames/s + the
# rocprofv3 kernel statistics of both), PMC passes of the default workload and of the configs[4] shard, the default bench line.
# usage: bash tools/profile_r06.sh <tag>   (outputs under gpurun_out/<tag>*; condensed into profiles/ by tools/make_profiles.py)
TAG=${1:-r06p}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/$TAG gpurun_out/${TAG}_old
timeout 400 python tools/ab_tune.py debug.fold_ln 0 1 > gpurun_out/${TAG}_ab_fold.txt 2>&1
timeout 300 python tools/ab_tune.py debug.fold_ln 0 1 --mode BBOX --skip-text > gpurun_out/${TAG}_ab_fold_bbox.txt 2>&1
timeout 300 python tools/ab_tune.py debug.fold_ln 0 1 --model L --template-size 256 --search-size 384 > gpurun_out/${TAG}_ab_fold_L.txt 2>&1
timeout 300 python tools/ab_tune.py fin_w 0 1 > gpurun_out/${TAG}_ab_finw.txt 2>&1
timeout 300 python tools/ab_tune.py fin_w 2 -1 > gpurun_out/${TAG}_ab_convfin.txt 2>&1
timeout 300 python tools/ab_tune.py debug.head_fin 0 1 > gpurun_out/${TAG}_ab_headfin.txt 2>&1
timeout 900 bash tools/profile_bench.sh gpurun_out/$TAG > gpurun_out/$TAG.log 2>&1
BENCH="python bench.py --steps 100 --warmup 20 --blocks 3 --no-cpu-baseline --no-batched --tune debug.fold_ln=0"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_old/stats -o bench --output-format csv -- $BENCH > gpurun_out/${TAG}_old/bench_stats_run.json 2> gpurun_out/${TAG}_old/stats.log
find gpurun_out/${TAG}_old -name "*kernel_trace.csv" -delete; find gpurun_out/${TAG}_old -name "*agent_info.csv" -delete
STEPS=30 WARMUP=5 timeout 1200 bash tools/profile_bench.sh gpurun_out/${TAG}_L8 --model L --batch 8 --template-size 256 --search-size 384 > gpurun_out/${TAG}_L8.log 2>&1
timeout 300 python bench.py --batch 8 --steps 60 --warmup 10 --no-cpu-baseline --no-batched > gpurun_out/${TAG}_bench_b8.json 2>/dev/null
timeout 300 python bench.py --batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-batched > gpurun_out/${TAG}_bench_b32.json 2>/dev/null
timeout 600 python bench.py > gpurun_out/${TAG}_bench_full.json 2>/dev/null
tail -3 gpurun_out/$TAG.log; cat gpurun_out/${TAG}_ab_fold.txt | tail -6
