"""Condense the rocprofv3 outputs a gpurun call left under gpurun_out/<tag>/ into the small, committed files of
profiles/: the --stats CSV, a per-kernel PMC summary (MFMA busy, HBM bytes) and the traffic table bench.py reads.
Usage: python tools/make_profiles.py <gpurun_out/tag> <round-prefix, e.g. r01> [model batch "workload text"]   (default: B 1)"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """'void uvl::gemm_glds_kernel<64, 64, 2, 2, 1, 3, false>(uvl::GemmParams)' -> 'gemm_glds_kernel<64,64,2,2,1,3,0>'"""
    n = name.split("(")[0].replace("void ", "").replace("uvl::", "").replace(" ", "")
    n = n.replace("false", "0").replace("true", "1")
    m = re.match(r"^(gemm_glds_kernel<\d+,\d+,\d+,\d+,\d+,\d+,\d+),(\d+)(?:,(\d+))?(?:,(\d+))?>$", n)
    if m:          # non-temporal weights / 32-wide K stages / producer waves: written the way the library names its launches (bench.py keys on that name)
        ntw, bk, prod = m.group(2), m.group(3) or "64", m.group(4) or "0"
        n = m.group(1) + (",nt>" if ntw == "1" else ",0,%s,%s>" % (bk, prod) if prod != "0" else ",0,32>" if bk == "32" else ">")
    return n


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        d[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return d, {k: len(v) for k, v in n.items()}


def main(src, tag, model="B", batch=1, wl="bench.py default workload: UVLTrack-B z256/x256/T40, batch 1"):
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    st = os.path.join(src, "stats", "bench_kernel_stats.csv")
    stats = {}
    if os.path.exists(st):
        shutil.copy(st, os.path.join(out, tag + "_bench_kernel_stats.csv"))
        for r in csv.DictReader(open(st)):
            stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
    lines = ["# %s -- rocprofv3 PMC summary per kernel (%s)" % (tag, wl), "",
             "Separate `--pmc` passes (MFMA / FETCH_SIZE / WRITE_SIZE), each with `--kernel-trace` only.  `avg us` is from the",
             "un-instrumented `--kernel-trace --stats` run (PMC passes serialise and slow the kernels).  HBM bytes per launch =",
             "2 x FETCH_SIZE + WRITE_SIZE (KB): MI355X_MICROARCH.md says FETCH_SIZE reports half of a wide coalesced stream.",
             "MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (avg duration x 2.1 GHz x 1024 SIMDs).", "",
             "| kernel | launches/run | avg us | % of GPU time | MFMA busy cyc/launch | MFMA util % | FETCH_SIZE KB | WRITE_SIZE KB | HBM MB/launch (corrected) |",
             "|---|---|---|---|---|---|---|---|---|"]
    traffic = {}
    m, nm = agg(os.path.join(src, "pmc_mfma", "bench_counter_collection.csv"))
    f, nf = agg(os.path.join(src, "pmc_fetch", "bench_counter_collection.csv"))
    w, nw = agg(os.path.join(src, "pmc_write", "bench_counter_collection.csv"))
    for k in sorted(stats, key=lambda k: -stats[k][2]):
        if not re.match(r"(gemm|attn|ln_|contrast|head|im2row|bert|setup|slab|prologue|conv_fin|text_join)", k):
            continue
        calls, avg_us, pct = stats[k]
        busy = m.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(nm.get(k, 1), 1)
        fe = f.get(k, {}).get("FETCH_SIZE", 0.0) / max(nf.get(k, 1), 1)
        wr = w.get(k, {}).get("WRITE_SIZE", 0.0) / max(nw.get(k, 1), 1)
        hbm = (2 * fe + wr) * 1024
        util = 100.0 * busy / (avg_us * 1e-6 * 2.1e9 * 1024) if avg_us > 0 else 0.0
        traffic[k] = hbm
        lines.append("| `%s` | %d | %.2f | %.1f | %.0f | %.1f | %.0f | %.0f | %.2f |" % (k, calls, avg_us, pct, busy, util, fe, wr, hbm / 1e6))
    open(os.path.join(out, tag + "_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py` (%s)" % tag,
               "workload": {"model": model, "batch": int(batch), "text": wl},
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch", "bytes_per_launch": traffic},
              open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1)
    for fn in os.listdir(src):
        if fn.startswith("bench_") and fn.endswith(".json") and "profile" not in fn:
            shutil.copy(os.path.join(src, fn), os.path.join(out, tag + "_" + fn))
    if os.path.exists(os.path.join(src, "attn_pmc.txt")):
        body = open(os.path.join(src, "attn_pmc.txt")).read()
        bench = open(os.path.join(src, "attn_bench.txt")).read() if os.path.exists(os.path.join(src, "attn_bench.txt")) else ""
        open(os.path.join(out, tag + "_attention_pmc.md"), "w").write(
            "# %s -- fused attention kernel alone (tools/attn_bench.py), rocprofv3 PMC passes on B=32, H=16, N=681\n\n"
            "Counters are per launch as rocprofv3 reports them.  Instruction mix: SQ_INSTS_VALU / SQ_INSTS_MFMA = VALU instructions per MFMA\n"
            "(16 MFMAs of 8 passes = 512 MFMA-pipe cycles per 64-key tile and wave); wave time: SQ_WAIT_ANY (s_waitcnt / barrier),\n"
            "SQ_WAIT_INST_ANY (issue stalls), SQ_ACTIVE_INST_ANY (issuing) as fractions of SQ_WAVE_CYCLES.\n\n```\n%s```\n\n"
            "Throughput of the same build on the frame's shapes:\n\n```\n%s```\n" % (tag, body, "\n".join(l for l in bench.splitlines() if l.startswith("attention")) + "\n"))
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:6])
