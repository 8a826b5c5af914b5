#!/usr/bin/env python3
"""tools/trace_summary.py -- per (kernel, grid size) durations from a rocprofv3 --kernel-trace CSV.

rocprofv3's own --stats table merges all dispatches of a kernel; bench.py launches k_striped on class C and,
for the scale_ref point, on class D -- since round 2 with the same grid (256 workgroups of 1024: one per CU), so
the summary is split by grid size AND, within one (kernel, grid), into groups of dispatches whose durations lie
more than a factor of three apart (class C ~72 us, class D ~1280 us).
    python tools/trace_summary.py <..._kernel_trace.csv>  > profiles/<name>.csv
"""
import csv
import re
import sys
from collections import defaultdict


def main():
    acc = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        m = re.match(r"[\w:]+(<[^(]*>)?", name)  # k_striped<double, 16, 2, 131072, 4>(...) -> up to the argument list
        name = m.group(0) if m else name
        acc[(name, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid_threads", "workgroup", "calls", "avg_us", "median_us", "min_us", "max_us", "total_ms"])
    rows = []
    groups = []
    for key, v in acc.items():
        v.sort()
        cur = [v[0]]
        for d in v[1:]:
            if d > 3.0 * cur[-1]:  # a gap of more than 3x: another workload on the same kernel and grid
                groups.append((key, cur))
                cur = []
            cur.append(d)
        groups.append((key, cur))
    for (name, grid, wg), v in groups:
        rows.append((sum(v), [name, grid, wg, len(v), "%.3f" % (sum(v) / len(v)), "%.3f" % v[len(v) // 2],
                              "%.3f" % v[0], "%.3f" % v[-1], "%.3f" % (sum(v) / 1000.0)]))
    for _, row in sorted(rows, reverse=True):
        w.writerow(row)


if __name__ == "__main__":
    main()
