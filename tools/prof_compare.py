"""Join several vcv_prof_dump CSVs (bench.py with VCVITS_PROF_DUMP) by launch shape and print ms/step per dump.
python tools/prof_compare.py steps name1=dump1.csv name2=dump2.csv ..."""
import sys
from collections import defaultdict
steps = int(sys.argv[1])
names, aggs = [], []
for arg in sys.argv[2:]:
    n, f = arg.split("=")
    names.append(n)
    agg = defaultdict(lambda: [0, 0.0, 0.0, set()])
    for l in open(f):
        r = l.strip().split(",")
        t = [int(v) for v in r[3:]]
        key = (int(r[0]), t[0], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9] % 10)
        a = agg[key]
        a[0] += 1; a[1] += float(r[1]); a[2] += float(r[2]); a[3].add((t[1], t[10], t[9] // 10))
    aggs.append(agg)
keys = sorted(set().union(*[set(a) for a in aggs]), key=lambda k: -max(a[k][1] if k in a else 0 for a in aggs))
print("cls   B   Cg   Mg  K     Q   P s ph am | calls " + " ".join("%9s" % n for n in names) + "  | TF/s " + " ".join("%7s" % n for n in names) + "  tiles")
tot = [0.0] * len(aggs)
best = 0.0
for k in keys:
    ms = [a[k][1] / steps if k in a else float("nan") for a in aggs]
    tf = [a[k][2] / a[k][1] if k in a and a[k][1] > 0 else float("nan") for a in aggs]
    calls = max(a[k][0] for a in aggs if k in a) / steps
    for i, m in enumerate(ms):
        if m == m:
            tot[i] += m
    best += min(m for m in ms if m == m)
    tiles = " ".join(sorted("%d:%dx%d/%d" % (e, tl // 1000, tl % 1000, ks) for a in aggs if k in a for (e, tl, ks) in a[k][3]))
    print("%3d %3d %4d %4d %2d %5d %3d %d %2d %2d | %5.1f " % (k + (calls,)) + " ".join("%9.3f" % m for m in ms) + "  |      " + " ".join("%7.1f" % t for t in tf) + "  " + tiles)
print("total ms/step: " + " ".join("%s %.2f" % (n, t) for n, t in zip(names, tot)) + "   best-of per shape %.2f" % best)
