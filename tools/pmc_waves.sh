OUT=gpurun_out/r02lds_L8
export TMPDIR=/tmp
BENCH="python bench.py --steps 10 --warmup 3 --blocks 3 --no-cpu-baseline --no-batched --model L --batch 8 --template-size 256 --search-size 384"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/act" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/act.log" || tail -3 "$OUT/act.log"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM -d "$OUT/act2" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/act2.log" || tail -3 "$OUT/act2.log"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d "$OUT/act3" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/act3.log" || tail -3 "$OUT/act3.log"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
python - "$OUT" <<'PY'
import collections, csv, re, sys
out = sys.argv[1]
def short(name):
    n = name.split("(")[0].replace("void ", "").replace("uvl::", "").replace(" ", "")
    return n.replace("false", "0").replace("true", "1")
for sub in ("act", "act2", "act3"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    try:
        for r in csv.DictReader(open("%s/%s/bench_counter_collection.csv" % (out, sub))):
            k = short(r["Kernel_Name"]); tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    except OSError as e:
        print("missing", sub, e); continue
    for k, c in tot.items():
        if not re.match(r"(gemm_glds_kernel<128|attn_stream)", k): continue
        wc = c.get("SQ_WAVE_CYCLES", 1)
        print(sub, k, len(n[k]), " ".join("%s=%.3g(%.1f%%)" % (a, v/len(n[k]), 100*v/wc) for a, v in sorted(c.items())))
PY
