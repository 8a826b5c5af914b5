#!/usr/bin/env python3
"""Kernel microbenchmark: times every plan / tuning combination on the benchmark shapes with HIP
events (spmv_hip_time_device) and prints achieved algorithmic GB/s.  GPU box only.

    python tools/kbench.py [--reps 50] [--cases C,sb120,rmat20,...]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import matgen  # noqa: E402
import pkgload  # noqa: E402


def build_case(pkg, name, dtype):
    if name in ("S", "W", "A", "B", "C"):
        m = pkg.npb.NpbMatrix(name)
        a = m.a if dtype == np.float64 else m.a.astype(np.float32)
        return m.rowstr, m.colidx, a, m  # keep m alive
    if name.startswith("sb"):
        return matgen.sparsebench_like(170, int(name[2:]), dtype) + (None,)
    if name.startswith("rmat"):
        return matgen.rmat_graph(20, int(name[4:]), 16, dtype, values="normal") + (None,)
    if name.startswith("grid"):
        n = int(name[4:])
        return matgen.grid2d_graph(n, n, dtype) + (None,)
    raise ValueError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--cases", default="C,sb100,rmat20,grid2000")
    ap.add_argument("--dtypes", default="f64")
    ap.add_argument("--tunings", default="xcd1nt0,xcd0nt0,xcd1nt1")
    args = ap.parse_args()
    pkg = pkgload.load_package()
    be = pkg.Backend()
    assert be.device_count() >= 1
    plans = [("stream", pkg.PLAN_STREAM, 0), ("vec64", pkg.PLAN_VECTOR, 64), ("vec32", pkg.PLAN_VECTOR, 32),
             ("vec16", pkg.PLAN_VECTOR, 16), ("vec8", pkg.PLAN_VECTOR, 8), ("vec4", pkg.PLAN_VECTOR, 4),
             ("vec1", pkg.PLAN_VECTOR, 1), ("binned", pkg.PLAN_BINNED, 0), ("ordered", pkg.PLAN_ORDERED, 0),
             ("auto", pkg.PLAN_AUTO, 0)]
    print("%-10s %-4s %-8s %-9s %10s %10s %8s" % ("case", "dt", "plan", "tuning", "us/spmv", "GB/s(alg)", "GFLOP/s"))
    for dts in args.dtypes.split(","):
        dtype = np.float64 if dts == "f64" else np.float32
        for case in args.cases.split(","):
            t0 = time.time()
            rs, ci, a, keep = build_case(pkg, case, dtype)
            rows = rs.size - 1
            dm = be.matrix(a, rs, ci, rows)
            inf = dm.info()
            x = np.random.default_rng(1).standard_normal(inf["cols"] + 2).astype(dtype)
            dx = be.malloc(x.nbytes)
            dy = be.malloc(rows * dtype().itemsize + 16)
            be.h2d(dx, x)
            print("# %s %s: rows=%d cols=%d nnz=%d len %d..%d mean %.1f blocks=%d long=%d alg_bytes=%d (setup %.1f s)" % (
                case, dts, rows, inf["cols"], inf["nnz"], inf["min_row_len"], inf["max_row_len"], inf["mean_row_len"],
                inf["n_blocks"], inf["n_long_rows"], inf["algorithmic_bytes"], time.time() - t0), flush=True)
            for tun in args.tunings.split(","):
                xcd, nt = int(tun[3]), int(tun[6])
                be.set_tuning(0, xcd)
                be.set_tuning(1, nt)
                for pname, plan, param in plans:
                    if tun != args.tunings.split(",")[0] and pname in ("ordered", "auto", "vec1", "vec4", "vec8"):
                        continue
                    dm.set_plan(plan, param)
                    reps = args.reps if pname not in ("ordered", "vec1") else max(3, args.reps // 10)
                    dm.time_device(dx, dy, 3)
                    ms = dm.time_device(dx, dy, reps) / reps
                    i2 = dm.info()
                    print("%-10s %-4s %-8s %-9s %10.2f %10.1f %8.1f   launches=%d" % (
                        case, dts, pname, tun, ms * 1e3, i2["algorithmic_bytes"] / ms / 1e6,
                        2.0 * i2["nnz"] / ms / 1e6, i2["n_launches"]), flush=True)
            be.set_tuning(0, 1)
            be.set_tuning(1, 0)
            dm.close()
            be.free(dx)
            be.free(dy)
            del keep


if __name__ == "__main__":
    main()
