"""Aggregate a vcv_prof_dump CSV by launch shape.  python tools/prof_by_shape.py dump.csv steps"""
import sys
from collections import defaultdict
rows = [l.strip().split(",") for l in open(sys.argv[1])]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    key = (int(r[0]),) + tuple(int(v) for v in r[3:])
    a = agg[key]
    a[0] += 1; a[1] += float(r[1]); a[2] += float(r[2])
tot = sum(a[1] for a in agg.values())
print("cls  B  G  Cg  Mg  K  Q/Ta  P  s  ph/Z amode/xs tile  BKC/NCH | calls/step  ms/step  TF/s  pct")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(" ".join("%5d" % k for k in key), "| %6.1f %8.3f %7.1f %5.1f%%" % (a[0] / steps, a[1] / steps, a[2] / a[1] if a[1] > 0 else 0, 100 * a[1] / tot))
print("total ms/step %.2f" % (tot / steps))
