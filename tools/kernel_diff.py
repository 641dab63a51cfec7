"""Per-kernel-family time of two rocprofv3 --kernel-trace --output-format csv runs, side by side (ms per step).
  python3 tools/kernel_diff.py DIR_A STEPS_A DIR_B STEPS_B [top]"""
import collections
import csv
import glob
import re
import sys


def load(d):
    tot = collections.Counter()
    n = collections.Counter()
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"<.*", "", r["Kernel_Name"]).split("(")[0].strip()
            name = name.replace("void ", "").replace("(anonymous namespace)::", "")
            dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            tot[name] += dt
            n[name] += 1
    return tot, n


def main():
    a, sa, b, sb = sys.argv[1], float(sys.argv[2]), sys.argv[3], float(sys.argv[4])
    top = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    ta, na = load(a)
    tb, nb = load(b)
    keys = sorted(set(ta) | set(tb), key=lambda k: -abs(ta[k] / sa - tb[k] / sb))
    print("# total ms/step: A %.2f   B %.2f   (all launches of the traced process / steps given)" % (sum(ta.values()) / sa, sum(tb.values()) / sb))
    print("%-52s %10s %8s %10s %8s %9s" % ("kernel", "A ms/step", "A calls", "B ms/step", "B calls", "B - A"))
    for k in keys[:top]:
        print("%-52s %10.3f %8.1f %10.3f %8.1f %+9.3f" % (k[:52], ta[k] / sa, na[k] / sa, tb[k] / sb, nb[k] / sb, tb[k] / sb - ta[k] / sa))


if __name__ == "__main__":
    main()
