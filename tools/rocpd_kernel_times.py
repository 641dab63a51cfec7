"""Kernel-only durations from a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace, grouped by (kernel, grid): calls and the
average duration -- for launches too short for host-side timing.  python tools/rocpd_kernel_times.py <results.db> [substring]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, count(*), avg(d.end - d.start) / 1000.0 "
         "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name, d.grid_size_x, d.grid_size_y order by 1, 2, 3" % (kd, ks))
    print("%-64s %22s %6s %10s" % ("kernel", "workgroups (x, y)", "calls", "avg_us"))
    for name, gx, gy, wx, n, us in db.execute(q):
        if pat and pat not in name:
            continue
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
        short = re.sub(r"EvP.*|Ev1.*|\.kd$", "", short)
        print("%-64s %12d x %7d %6d %10.1f" % (short[:64], gx // max(wx, 1), gy, n, us))


if __name__ == "__main__":
    main()
