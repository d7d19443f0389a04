"""Summarise a rocprofv3 results.db (kernel trace) into a per-kernel table: calls, total/avg/min/max duration.
Usage: python tools/rocprof_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def summarise(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, t, a, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + "..."
        out.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (short, c, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))
    return "\n".join(out)


if __name__ == "__main__":
    s = summarise(sys.argv[1])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(s + "\n")
    print(s)
