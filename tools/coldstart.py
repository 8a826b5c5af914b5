#!/usr/bin/env python3
"""What the first launches over a freshly created STRIPED layout cost.  Run under
`rocprofv3 --kernel-trace --output-format csv`: creates NPB class C with the host builder, then with the device builder,
then class D, and launches each 300 times back to back; tools/coldstart.py --summarise <kernel_trace.csv> prints the mean
k_striped duration by launch number.  GPU box only."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import pkgload
    pkg = pkgload.load_package()
    be = pkg.Backend()
    for cls, builder in (("C", 1), ("C", 2), ("C", 1), ("C", 2), ("D", 2)):
        m = pkg.npb.NpbMatrix(cls)
        be.set_tuning(17, builder)
        dm = be.matrix(m.a, m.rowstr, m.colidx, m.rows)
        x = np.random.default_rng(1).standard_normal(m.na + 2)
        dx, dy = be.malloc(x.nbytes), be.malloc(m.rows * 8 + 16)
        be.h2d(dx, x)
        ms = dm.time_device(dx, dy, 300)
        print("class %s builder %s: 300 launches, %.2f us each on average" % (cls, "host" if builder == 1 else "device", ms / 300 * 1e3), flush=True)
        dm.close()
        be.free(dx), be.free(dy)
        m.close()
    be.set_tuning(17, 0)


def summarise(path):
    rows = [r for r in csv.DictReader(open(path)) if "k_striped" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    runs = [rows[i:i + 300] for i in range(0, len(rows) - 299, 300)]
    names = ["class C, host-built layout", "class C, device-built layout", "class C, host-built (again)", "class C, device-built (again)",
             "class D, device-built layout"]
    print("# mean k_striped duration (us) by launch number after the layout was created; rocprofv3 kernel trace")
    print("%-34s %8s %8s %8s %8s %8s %8s" % ("", "1", "2-5", "6-20", "21-50", "51-100", "101-300"))
    for name, run in zip(names, runs):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in run]
        seg = lambda a, b: sum(d[a:b]) / (b - a)  # noqa: E731
        print("%-34s %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f" % (name, d[0], seg(1, 5), seg(5, 20), seg(20, 50), seg(50, 100), seg(100, 300)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run()
