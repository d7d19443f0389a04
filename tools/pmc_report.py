"""Aggregate rocprofv3 --pmc counter_collection CSVs (one or more passes) into per-kernel totals per launch.
Usage: python tools/pmc_report.py <dir-with-pass-subdirs> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys


def main(root, filt=""):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uvl::", "")
            if filt and filt not in k:
                continue
            d[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in sorted(d):
        print(k)
        for c in sorted(d[k]):
            cnt = len(n[k][c])
            print("    %-34s %16.0f per launch (%d launches)" % (c, d[k][c] / cnt, cnt))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
