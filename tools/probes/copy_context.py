"""Where the runtime's blit kernels (__amd_rocclr_copyBuffer / fillBuffer) of a step come from: for each one in a rocprofv3
kernel trace (rocpd sqlite) the kernels issued just before and after it on the same queue, counted.
  python3 tools/probes/copy_context.py <results.db> [name substring, default copyBuffer]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"void |\(anonymous namespace\)::|at::native::", "", n)
    return re.sub(r"[<(].*", "", n)[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute("select s.kernel_name, d.grid_size_x, d.start, d.end from %s d join %s s on d.kernel_id = s.id "
                           "order by d.start" % (kd, ks)))
    ctx = collections.Counter()
    dur = collections.Counter()
    for i, (n, gx, st, en) in enumerate(rows):
        if pat in n:
            prev = short(rows[i - 1][0]) if i else "-"
            nxt = short(rows[i + 1][0]) if i + 1 < len(rows) else "-"
            ctx[(prev, gx, nxt)] += 1
            dur[(prev, gx, nxt)] += (en - st) / 1000.0
    print("%-42s %10s %-42s %6s %8s" % ("previous kernel", "grid_x", "next kernel", "count", "avg_us"))
    for k, c in ctx.most_common(40):
        print("%-42s %10d %-42s %6d %8.1f" % (k[0], k[1], k[2], c, dur[k] / c))


if __name__ == "__main__":
    main()
