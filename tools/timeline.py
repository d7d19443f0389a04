"""Timeline of one steady-state frame from a rocprofv3 kernel trace (results.db): per-queue busy time, gaps,
and the kernel list.  Usage: python tools/timeline.py <results.db> [n_rows]"""
import sqlite3
import sys


def main(path, nrows=200):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "setup_kernel" in r[0]]
    a, b = idx[-3], idx[-2]
    frame = rows[a:b]
    t0 = frame[0][1]
    print("frame span %.1f us, %d kernels, busy sum %.1f us" % ((rows[b][1] - t0) / 1e3, len(frame), sum(r[2] - r[1] for r in frame) / 1e3))
    qs = {}
    for r in frame:
        qs.setdefault(r[3], []).append(r)
    for q, v in qs.items():
        gaps = [max(0, v[i + 1][1] - v[i][2]) for i in range(len(v) - 1)]
        print(" queue %s: n=%d busy %.1f us, first start %.1f, last end %.1f, gaps sum %.1f max %.1f" % (
            q, len(v), sum(r[2] - r[1] for r in v) / 1e3, (v[0][1] - t0) / 1e3, (v[-1][2] - t0) / 1e3, sum(gaps) / 1e3, max(gaps) / 1e3 if gaps else 0))
    agg = {}
    for r in frame:
        k = r[0].split("(")[0][-60:]
        a_ = agg.setdefault(k, [0, 0.0])
        a_[0] += 1
        a_[1] += (r[2] - r[1]) / 1e3
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("   %-62s n=%3d total %7.1f us avg %6.2f" % (k, v[0], v[1], v[1] / v[0]))
    for r in frame[:nrows]:
        print("%8.1f %8.1f %6.1f q%s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0][:70]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200)
