#!/usr/bin/env python3
"""What bounds k_stream on the graph shapes?  Timing-only ablations of the debug library's instrumented twin (tuning key
6; spmv_kernels.hip k_stream<DIAG>), interleaved with the real kernel over several rounds:
    1  no phase 2 (the lane groups that reduce rows out of LDS): loads + gathers + LDS parking alone
    2  every gather goes to the first KiB of x: the gather MISSES taken away
    3  both: the streaming skeleton (a, colidx, row pointers read once)
    7  the skeleton without the row pointers
k_long / k_long_finish run unchanged behind every variant (they are part of the STREAM plan's product).
GPU box only.   python tools/streamabl.py [--cases rmat20,rmat22] [--rounds 6]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import pkgload  # noqa: E402
from kbench import build_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="rmat20,rmat22")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--masks", default="0,1,2,3,7")
    ap.add_argument("--plan", default="stream")
    args = ap.parse_args()
    pkg = pkgload.load_package()
    be = pkg.Backend(pkg.lib_path("libhip-spmv-debug.so"))
    masks = [int(v) for v in args.masks.split(",")]
    plan = {v: k for k, v in pkg.abi.PLAN_NAMES.items()}[args.plan]
    for case in args.cases.split(","):
        rs, ci, a, keep = build_case(pkg, case, np.float64)
        rows = rs.size - 1
        dm = be.matrix(a, rs, ci, rows)
        inf = dm.info()
        x = np.random.default_rng(1).standard_normal(inf["cols"] + 2)
        dx, dy = be.malloc(x.nbytes), be.malloc(rows * 8 + 16)
        be.h2d(dx, x)
        dm.set_plan(plan, 0)
        i2 = dm.info()
        print("# %s: rows=%d nnz=%d empty rows=%d row blocks=%d long rows=%d launches=%d alg_bytes=%d" % (
            case, rows, inf["nnz"], int(np.count_nonzero(np.diff(rs) == 0)), i2["n_blocks"], i2["n_long_rows"], i2["n_launches"],
            i2["algorithmic_bytes"]), flush=True)
        acc = {mk: [] for mk in masks}
        for _ in range(args.rounds):
            for mk in masks:
                be.set_tuning(6, mk)
                dm.time_device(dx, dy, 3)
                acc[mk].append(dm.time_device(dx, dy, args.reps) / args.reps * 1e3)
        be.set_tuning(6, 0)
        for mk in masks:
            v = np.array(acc[mk])
            print("%-8s mask %d: mean %8.2f us  median %8.2f  min %8.2f   (%.0f GB/s algorithmic)" % (
                case, mk, v.mean(), np.median(v), v.min(), i2["algorithmic_bytes"] / np.median(v) / 1e3), flush=True)
        dm.close()
        be.free(dx)
        be.free(dy)


if __name__ == "__main__":
    main()
