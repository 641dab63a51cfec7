"""MFMA-pipe-busy summary of a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass.
python tools/pmc_busy_summary.py <dir with *_counter_collection.csv> <out.txt> "<command line profiled>"
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs) / (256 CUs x 4 SIMDs): both counters are sums over the chip
(MI355X_MICROARCH.md, rocprofv3 PMC section)."""
import collections
import csv
import glob
import re
import sys

src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
files = glob.glob(src + "/*counter_collection.csv") + glob.glob(src + "/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(files[0])):
    m = re.search(r"((?:conv|wgrad|rel_attn|grouped|resblock)\w*_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
    if not m:
        continue
    n = m.group(1)
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[n] += 1
with open(dst, "w") as o:
    o.write("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- %s\n" % cmd)
    o.write("# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8) / 1024   (counter sums over the chip)\n")
    o.write("%-64s %8s %14s %14s %8s\n" % ("kernel", "launches", "MFMA_BUSY", "GUI_ACTIVE", "busy"))
    for n, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]):
        o.write("%-64s %8d %14.4g %14.4g %8.4f\n" % (n[:64], cnt[n], c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"],
                                                       c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c["GRBM_GUI_ACTIVE"], 1) * 8 / 1024))
print(open(dst).read())
