# end-of-round evidence run (GPU box, repo root): gpurun -- "bash tools/final_run.sh"; then tools/make_profiles.py condenses gpurun_out/r05_prof, r05_L8_prof into profiles/: tests, smoke, bench lines, rocprofv3 profiles
python -m pytest tests -m gpu -x -q > gpurun_out/t_final.log 2>&1; tail -3 gpurun_out/t_final.log
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-200 gpurun_out/bench_final.json
python bench.py --batch 8 --no-cpu-baseline --no-batched > gpurun_out/bench_b8.json 2>/dev/null
python bench.py --batch 32 --no-cpu-baseline --no-batched > gpurun_out/bench_b32.json 2>/dev/null
python bench.py --model L --batch 32 --steps 60 --warmup 10 --no-cpu-baseline --no-batched > gpurun_out/bench_L32.json 2>/dev/null
STEPS=100 WARMUP=20 bash tools/profile_bench.sh gpurun_out/r05_prof > /dev/null 2>&1
STEPS=30 WARMUP=8 bash tools/profile_bench.sh gpurun_out/r05_L8_prof --model L --batch 8 > /dev/null 2>&1
ls gpurun_out/r05_prof gpurun_out/r05_L8_prof | head -30
python tools/lib_compare.py cfg4 > gpurun_out/r05_lib_compare_cfg4.txt 2>&1
python tools/attn_bench.py > gpurun_out/r05_attn_bench.txt 2>&1
python tools/res_epilogue_probe.py > gpurun_out/r05_res_epilogue.txt 2>&1
python bench.py --gpus 1 --dist --model L --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-batched > gpurun_out/bench_L8_dist.json 2>/dev/null
python bench.py --gpus 1 --dist --steps 100 --warmup 20 --no-cpu-baseline --no-batched > gpurun_out/bench_b1_dist.json 2>/dev/null
