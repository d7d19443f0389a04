python bench.py --model L --batch 8 --template-size 256 --search-size 384 --steps 30 --warmup 5 --no-cpu-baseline --no-batched --profile-json gpurun_out/r2_prof_L8.json > gpurun_out/r2_bench_L8.json
python bench.py --batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-batched --profile-json gpurun_out/r2_prof_b32.json > gpurun_out/r2_bench_b32.json
python - <<'PY'
import json
for f in ("gpurun_out/r2_prof_L8.json","gpurun_out/r2_prof_b32.json"):
    d=json.load(open(f)); tot=sum(v["ms"] for v in d["by_kernel"].values())
    print(f, "fps", d["line"]["value"], "sum ms", tot)
    for k,v in sorted(d["by_kernel"].items(), key=lambda kv:-kv[1]["ms"])[:14]:
        print("  %-44s %7.3f ms %5.1f%% n=%3d  %7.1f TF  %s" % (k, v["ms"], 100*v["ms"]/tot, v["launches"], v["flops"]/v["ms"]/1e9 if v["ms"] else 0, ",".join(sorted(set(v["sites"])))[:50]))
PY
