#!/bin/bash
# Collect the evidence committed under profiles/ for one round.  Run on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh r01'    then here:   python tools/make_profiles.py gpurun_out/r01 r01
# Passes (PMC counters in their own runs, kernel trace only, as the profiling guide requires):
#   stats      rocprofv3 --kernel-trace --stats               python bench.py --steps 100 --warmup 20
#   pmc_mfma   --pmc SQ_VALU_MFMA_BUSY_CYCLES ...             python bench.py --steps 4 --warmup 2
#   pmc_fetch  --pmc FETCH_SIZE        pmc_write  --pmc WRITE_SIZE
# plus un-instrumented bench lines for the default workload, batch 8 / 32 and UVLTrack-L.
set -u
TAG=${1:-r01}
REPO=$(pwd)
O=$REPO/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --steps 100 --warmup 20 --no-cpu-baseline > $O/stats_run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma -o bench -- $B --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $REPO
python bench.py --steps 300 --warmup 50 --profile-json $O/bench_profile.json > $O/bench_default.json 2> $O/bench_default.err
python bench.py --batch 2 --steps 200 --warmup 30 --no-cpu-baseline > $O/bench_b2.json 2>/dev/null
python bench.py --batch 4 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_b4.json 2>/dev/null
python bench.py --batch 8 --steps 60 --warmup 10 --no-cpu-baseline --profile-json $O/bench_profile_b8.json > $O/bench_b8.json 2>/dev/null
python bench.py --batch 32 --steps 30 --warmup 5 --no-cpu-baseline --profile-json $O/bench_profile_b32.json > $O/bench_b32.json 2>/dev/null
python bench.py --model L --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_L.json 2>/dev/null
python bench.py --model L --batch 8 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_L_b8.json 2>/dev/null
python bench.py --mode BBOX --steps 300 --warmup 50 --no-cpu-baseline > $O/bench_bbox.json 2>/dev/null
python bench.py --mode BBOX --skip-text --steps 300 --warmup 50 --no-cpu-baseline > $O/bench_bbox_notext.json 2>/dev/null
python bench.py --reuse-text --steps 300 --warmup 50 --no-cpu-baseline > $O/bench_reuse_text.json 2>/dev/null
python bench.py --reuse-text --batch 8 --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_reuse_text_b8.json 2>/dev/null
python bench.py --mode NL --steps 300 --warmup 50 --no-cpu-baseline > $O/bench_nl.json 2>/dev/null
python bench.py --template-size 128 --steps 300 --warmup 50 --no-cpu-baseline > $O/bench_z128.json 2>/dev/null
python bench.py --model L --template-size 128 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_L_z128.json 2>/dev/null
# fused-attention kernel alone on the batched UVLTrack-L shape (B=32, H=16, N=681): instruction mix and busy counters
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/attn/p1 -o a -- python $REPO/tools/attn_bench.py 32 16 681 > $O/attn_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/attn/p2 -o a -- python $REPO/tools/attn_bench.py 32 16 681 > $O/attn_p2.log 2>&1
cd $REPO
python tools/pmc_report.py $O/attn attn_kernel > $O/attn_pmc.txt 2>&1
python tools/attn_bench.py > $O/attn_bench.txt 2>&1
for f in default b2 b4 b8 b32 L L_b8 bbox bbox_notext reuse_text reuse_text_b8 nl z128 L_z128; do tail -1 $O/bench_$f.json | cut -c1-140; done
ls $O/stats $O/pmc_mfma | head -20
