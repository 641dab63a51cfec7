"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / average
duration, like `--stats` prints.  Usage: python tools/rocpd_summary.py <results.db> [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute("select %s, start, end from kernels" % name_col))
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values())
    t0 = min(r[1] for r in rows)
    t1 = max(r[2] for r in rows)
    lines = ["# kernels: %d dispatches, %.3f ms busy of %.3f ms trace span" % (len(rows), total / 1e6, (t1 - t0) / 1e6),
             "%-112s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct")]
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%-112s %8d %12.3f %10.2f %6.2f%%" % (short(n), c, d / 1e6, d / c / 1e3, 100.0 * d / total))
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
