#!/bin/bash
# LDS utilisation of the batched workload's kernels (GPU box, repo root): separate --pmc passes, --kernel-trace only.
# usage: bash tools/pmc_lds.sh gpurun_out/<tag> [bench.py flags]
OUT=${1:-gpurun_out/pmc_lds}; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
BENCH="python bench.py --steps 10 --warmup 3 --blocks 3 --no-cpu-baseline --no-batched $*"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d "$OUT/lds" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/lds.log" || tail -3 "$OUT/lds.log"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d "$OUT/wait" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/wait.log" || tail -3 "$OUT/wait.log"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU -d "$OUT/insts" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/insts.log" || tail -3 "$OUT/insts.log"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
python - "$OUT" <<'PY'
import collections, csv, re, sys
out = sys.argv[1]
def short(name):
    n = name.split("(")[0].replace("void ", "").replace("uvl::", "").replace(" ", "")
    return n.replace("false", "0").replace("true", "1")
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for sub in ("lds", "wait", "insts"):
    try:
        for r in csv.DictReader(open("%s/%s/bench_counter_collection.csv" % (out, sub))):
            k = short(r["Kernel_Name"]); tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, sub)].add(r["Dispatch_Id"])
    except OSError as e:
        print("missing", sub, e)
print("| kernel | launches | LDS active % of CU-cycles | bank-conflict % of LDS active | wait-any % of wave cycles | LDS-issue wait % | LDS inst / MFMA | VALU / MFMA | SALU / MFMA |")
print("|---|---|---|---|---|---|---|---|---|")
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if not re.match(r"(gemm|attn|ln_)", k): continue
    n = len(cnt[(k, "lds")]) or 1
    gui = c.get("GRBM_GUI_ACTIVE", 0)
    act = c.get("SQ_LDS_IDX_ACTIVE", 0); bc = c.get("SQ_LDS_BANK_CONFLICT", 0)
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1; mf = c.get("SQ_INSTS_MFMA", 0) or 1
    print("| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.2f | %.1f | %.1f |" % (k, n, 100 * act / (gui * 256) if gui else 0, 100 * bc / act if act else 0,
          100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_INSTS_LDS", 0) / mf, c.get("SQ_INSTS_VALU", 0) / mf, c.get("SQ_INSTS_SALU", 0) / mf))
PY
