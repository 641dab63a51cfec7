"""Per-kernel-family averages of a rocprofv3 --pmc counter_collection CSV.  python tools/pmc_kernel.py <csv> [substr]"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if sub and sub not in n:
        continue
    m = re.search(r"([a-z0-9_]+_kernel(<[^>]*>)?)", n)
    k = m.group(1) if m else n[:50]
    a = agg[k][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print("   %-28s launches %5d  avg %16.1f" % (c, n, v / n))
