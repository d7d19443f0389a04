"""Per-queue timeline of one steady-state frame from a rocprofv3 kernel trace CSV (`--kernel-trace --output-format csv`).
Usage: python tools/frame_timeline.py <kernel_trace.csv> [frame-index-from-end]"""
import csv
import sys


def short(n):
    return n.split("(")[0].replace("void ", "").replace("uvl::", "").replace(" ", "")[:44]


def main(path, back=3):
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "im2row" in r["Kernel_Name"]]
    if len(starts) < back + 1:
        raise SystemExit("not enough frames in the trace")
    a, b = starts[-back - 1], starts[-back]
    # the text branch of a frame may start before its im2row: take kernels between the two frame markers by time
    t0 = int(rows[a]["Start_Timestamp"])
    frame = [r for r in rows[a - 4:b] if int(r["Start_Timestamp"]) >= t0 - 20000 and int(r["Start_Timestamp"]) < int(rows[b]["Start_Timestamp"])]
    queues = sorted({r["Queue_Id"] for r in frame})
    print("frame of %d kernels, %.1f us from first start to last end; queues %s" % (
        len(frame), (max(int(r["End_Timestamp"]) for r in frame) - t0) / 1e3, queues))
    for q in queues:
        ks = [r for r in frame if r["Queue_Id"] == q]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks) / 1e3
        print("queue %s: %3d kernels, busy %.1f us, first start %+.1f us, last end %+.1f us" % (
            q, len(ks), busy, (int(ks[0]["Start_Timestamp"]) - t0) / 1e3, (max(int(r["End_Timestamp"]) for r in ks) - t0) / 1e3))
    print()
    for r in frame:
        print("q%s %+8.1f .. %+8.1f  (%5.1f us)  %s" % (r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, short(r["Kernel_Name"])))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
