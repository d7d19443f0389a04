#!/bin/bash
# Round-end evidence run (GPU box, repo root): rocprofv3 stats + PMC passes of bench.py's default and batched workloads, PMC passes and
# the throughput table of the fused attention kernel, the library-comparison table and the batch 8 / 32 bench lines.
# usage: bash tools/profile_final.sh <tag>   (outputs under gpurun_out/<tag>*; condensed into profiles/ by tools/make_profiles.py)
TAG=${1:-r02p}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
bash tools/profile_bench.sh gpurun_out/$TAG > gpurun_out/$TAG.log 2>&1
STEPS=30 WARMUP=5 bash tools/profile_bench.sh gpurun_out/${TAG}_L8 --model L --batch 8 --template-size 256 --search-size 384 > gpurun_out/${TAG}_L8.log 2>&1
bash tools/pmc_attn.sh gpurun_out/${TAG}_attn 32 16 681 -1 > gpurun_out/${TAG}_attn.log 2>&1
bash tools/pmc_attn.sh gpurun_out/${TAG}_attn873 8 16 873 -1 > gpurun_out/${TAG}_attn873.log 2>&1
python tools/attn_bench.py --cfgs 8,10,11 > gpurun_out/${TAG}_attn_bench.txt 2>&1
python tools/lib_compare.py 8 32 > gpurun_out/${TAG}_lib_compare.txt 2>&1
python bench.py --batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-batched > gpurun_out/${TAG}_bench_b32.json 2>/dev/null
python bench.py --batch 8 --steps 60 --warmup 10 --no-cpu-baseline --no-batched > gpurun_out/${TAG}_bench_b8.json 2>/dev/null
python bench.py > gpurun_out/${TAG}_bench_full.json 2>/dev/null
tail -3 gpurun_out/$TAG.log; tail -3 gpurun_out/${TAG}_attn.log; cut -c1-200 gpurun_out/${TAG}_bench_b32.json
