"""Turn the rocprofv3 CSV outputs of one bench run into the committed summaries under profiles/.

  python tools/profile_summary.py gpurun_out/r1 profiles/r1

Inputs (produced on the GPU box, see profiles/README.md for the exact commands):
  <dir>/trace/t_kernel_stats.csv        rocprofv3 --kernel-trace --stats
  <dir>/fetch/f_counter_collection.csv  rocprofv3 --pmc FETCH_SIZE   (own pass)
  <dir>/write/w_counter_collection.csv  rocprofv3 --pmc WRITE_SIZE   (own pass)
HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read stream, so the read side is
given both raw and doubled (the dword-per-lane staging loads of these kernels are not one of the
calibrated access shapes -- the doubled figure is an upper bound)."""
import collections
import csv
import os
import re
import sys


def family(name):
    m = re.search(r"(conv_gemm_kernel|conv_wgrad_kernel|[a-z0-9_]+_kernel|vectorized_elementwise_kernel|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:40]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    rows = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst + "_kernel_stats.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline\n")
        f.write("# total kernel time %.3f ms over 4 steps (1 warm-up + 3 timed)\n" % (tot / 1e6))
        f.write("%-100s %7s %12s %11s %7s\n" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
        for r in rows[:45]:
            n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")
            f.write("%-100s %7d %12.3f %11.2f %6.2f%%\n" % (n[:100], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                            float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        fam = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            a = fam[family(r["Name"])]
            a[0] += int(r["Calls"])
            a[1] += float(r["TotalDurationNs"])
        f.write("\n# by kernel family\n")
        for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:12]:
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
    with open(dst + "_hbm_traffic.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1\n")
        f.write("# values in KiB as reported; read side also shown x2 (gfx950 FETCH_SIZE correction, upper bound here)\n")
        f.write("%-28s %8s %14s %14s %14s %16s\n" % ("kernel family", "launches", "fetch_KiB/launch", "x2", "write_KiB/launch", "MB/launch (x2+w)"))
        fams = sorted(traffic.get("FETCH_SIZE", {}), key=lambda k: -traffic["FETCH_SIZE"][k][1])[:10]
        for k in fams:
            c, v = traffic["FETCH_SIZE"][k]
            wc, wv = traffic.get("WRITE_SIZE", {}).get(k, [1, 0.0])
            f.write("%-28s %8d %14.1f %14.1f %14.1f %16.3f\n" % (k, c, v / c, 2 * v / c, wv / max(wc, 1),
                                                                 (2 * v / c + wv / max(wc, 1)) * 1024 / 1e6))
    # machine-readable per-launch traffic of the dominant kernel for bench.py's roofline.traffic
    import json
    if "FETCH_SIZE" in traffic and "conv_dma_kernel" in traffic["FETCH_SIZE"]:
        c, v = traffic["FETCH_SIZE"]["conv_dma_kernel"]
        wc, wv = traffic.get("WRITE_SIZE", {}).get("conv_dma_kernel", [1, 0.0])
        json.dump({"kernel": "conv_dma_kernel", "launches_profiled": c,
                   "fetch_bytes_per_launch_raw": v / c * 1024, "fetch_bytes_per_launch_x2": 2 * v / c * 1024,
                   "write_bytes_per_launch": wv / max(wc, 1) * 1024,
                   "hbm_bytes_per_launch": (2 * v / c + wv / max(wc, 1)) * 1024,
                   "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); "
                           "upper bound for these dword-per-lane staging loads"}, open(dst + "_hbm_traffic.json", "w"), indent=1)
    print(open(dst + "_kernel_stats.txt").read()[-1500:])
    print(open(dst + "_hbm_traffic.txt").read())


if __name__ == "__main__":
    main()
