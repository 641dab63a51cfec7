"""Turn the rocprofv3 CSV outputs of one bench run into the committed summaries under profiles/.

  python tools/profile_summary.py gpurun_out/r2 profiles/r2 [workload-key]

Inputs (produced on the GPU box, see profiles/README.md for the exact commands):
  <dir>/trace/t_kernel_stats.csv        rocprofv3 --kernel-trace --stats
  <dir>/fetch/f_counter_collection.csv  rocprofv3 --pmc FETCH_SIZE   (own pass)
  <dir>/write/w_counter_collection.csv  rocprofv3 --pmc WRITE_SIZE   (own pass)
HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced read stream, so the read side is given both raw and doubled (validated
on adamw_kernel, whose byte count is known exactly).  Outputs:
  <dst>_kernel_stats.txt   per-kernel and per-family time
  <dst>_hbm_traffic.txt    bytes per launch per family
  <dst>_hbm_rates.txt      achieved HBM GB/s per family = (FETCH x2 + WRITE) / average launch duration, vs 8 TB/s
  <dst>_hbm_traffic.json   machine-readable, stamped with the kernel-source hash and the workload key bench.py checks
"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def family(name):
    m = re.search(r"(conv_gemm_kernel|conv_wgrad_kernel|[a-z0-9_]+_kernel|vectorized_elementwise_kernel|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:40]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "base/vocoder/f32"
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    rows = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = fam[family(r["Name"])]
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    with open(dst + "_kernel_stats.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py ... (workload %s)\n" % workload)
        f.write("# total kernel time %.3f ms\n" % (tot / 1e6))
        f.write("%-100s %7s %12s %11s %7s\n" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
        for r in rows[:45]:
            n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")
            f.write("%-100s %7d %12.3f %11.2f %6.2f%%\n" % (n[:100], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                            float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        f.write("\n# by kernel family\n")
        for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
            f.write("%-40s calls %7d  total %10.3f ms  avg %9.2f us  %5.1f%%\n" % (k, c, d / 1e6, d / c / 1e3, 100 * d / tot))
    traffic = {}
    for tag, sub, fn in (("FETCH_SIZE", "fetch", "f_counter_collection.csv"), ("WRITE_SIZE", "write", "w_counter_collection.csv")):
        p = os.path.join(src, sub, fn)
        if not os.path.exists(p):
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] != tag:
                continue
            a = agg[family(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        traffic[tag] = agg
    fams = sorted(traffic.get("FETCH_SIZE", {}), key=lambda k: -traffic["FETCH_SIZE"][k][1])
    with open(dst + "_hbm_traffic.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)\n")
        f.write("# values in KiB as reported; read side also shown x2 (gfx950 FETCH_SIZE correction)\n")
        f.write("%-32s %8s %14s %14s %14s %16s\n" % ("kernel family", "launches", "fetch_KiB/launch", "x2", "write_KiB/launch", "MB/launch (x2+w)"))
        for k in fams[:14]:
            c, v = traffic["FETCH_SIZE"][k]
            wc, wv = traffic.get("WRITE_SIZE", {}).get(k, [1, 0.0])
            f.write("%-32s %8d %14.1f %14.1f %14.1f %16.3f\n" % (k, c, v / c, 2 * v / c, wv / max(wc, 1),
                                                                 (2 * v / c + wv / max(wc, 1)) * 1024 / 1e6))
    kern = {}
    with open(dst + "_hbm_rates.txt", "w") as f:
        f.write("# achieved HBM rate per kernel family: (FETCH_SIZE x2 + WRITE_SIZE) per launch / average launch duration of the\n")
        f.write("# --kernel-trace --stats pass of the same command; peak 8 TB/s (MI355X_MICROARCH.md; ~6.3 TB/s is the measured copy rate)\n")
        f.write("%-32s %8s %12s %10s %10s %9s\n" % ("kernel family", "launches", "MB/launch", "avg_us", "GB/s", "of 8TB/s"))
        for k in fams:
            c, v = traffic["FETCH_SIZE"][k]
            wc, wv = traffic.get("WRITE_SIZE", {}).get(k, [1, 0.0])
            by = (2 * v / c + wv / max(wc, 1)) * 1024
            if k not in fam or fam[k][0] == 0:
                continue
            us = fam[k][1] / fam[k][0] / 1e3
            gbs = by / (us * 1e-6) / 1e9
            kern[k] = {"launches_profiled": c, "fetch_bytes_per_launch_raw": v / c * 1024,
                       "fetch_bytes_per_launch_x2": 2 * v / c * 1024, "write_bytes_per_launch": wv / max(wc, 1) * 1024,
                       "hbm_bytes_per_launch": by, "avg_launch_us": us, "hbm_GBps": gbs}
            f.write("%-32s %8d %12.3f %10.2f %10.1f %8.3f\n" % (k, c, by / 1e6, us, gbs, gbs / 8000.0))
    from bench import kernel_source_hash
    json.dump({"workload": workload, "kernel_source_hash": kernel_source_hash(), "kernels": kern,
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)"},
              open(dst + "_hbm_traffic.json", "w"), indent=1)
    print(open(dst + "_kernel_stats.txt").read()[-2000:])
    print(open(dst + "_hbm_rates.txt").read())


if __name__ == "__main__":
    main()
