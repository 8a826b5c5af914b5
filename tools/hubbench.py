#!/usr/bin/env python3
"""The HUB plan by size of its second-tier table (tuning key 22; the first tier is the LDS table, 16 352 doubles): one layout
per size on the same matrix, STREAM beside it, interleaved over rounds.  GPU box only.
    python tools/hubbench.py [--cases rmat20,rmat22] [--tables 0,524288,1048576,2097152,4194304] [--rounds 3] [--dtypes f64]"""
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
    ap.add_argument("--tables", default="0,524288,1048576,2097152,4194304")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtypes", default="f64")
    args = ap.parse_args()
    pkg = pkgload.load_package()
    be = pkg.Backend()
    tables = [int(v) for v in args.tables.split(",")]
    for dts in args.dtypes.split(","):
        dtype = np.float64 if dts == "f64" else np.float32
        for case in args.cases.split(","):
            rs, ci, a, keep = build_case(pkg, case, dtype)
            rows = rs.size - 1
            cnt = np.sort(np.bincount(ci - 1))[::-1].cumsum()
            ms = []
            for tb in tables:
                be.set_tuning(22, tb)
                m = be.matrix(a, rs, ci, rows)
                m.set_plan(pkg.PLAN_HUB, 0)
                ms.append(m)
            be.set_tuning(22, 2 << 20)
            m0 = be.matrix(a, rs, ci, rows)
            m0.set_plan(pkg.PLAN_STREAM, 0)
            inf = m0.info()
            x = np.random.default_rng(1).standard_normal(inf["cols"] + 2).astype(dtype)
            dx, dy = be.malloc(x.nbytes), be.malloc(rows * x.itemsize + 16)
            be.h2d(dx, x)
            y0 = None
            print("# %s %s: rows=%d cols=%d nnz=%d alg_bytes=%d" % (case, dts, rows, inf["cols"], inf["nnz"], inf["algorithmic_bytes"]), flush=True)
            acc = {i: [] for i in range(len(ms) + 1)}
            for _ in range(args.rounds):
                for i, m in enumerate(ms + [m0]):
                    m.time_device(dx, dy, 3)
                    acc[i].append(m.time_device(dx, dy, args.reps) / args.reps * 1e3)
            for i, m in enumerate(ms):
                K = m.info()["plan_param"]
                k2 = tables[i] // x.itemsize
                cov0 = cnt[min(K, cnt.size) - 1] / inf["nnz"] if K else 0.0
                cov1 = cnt[min(K + k2, cnt.size) - 1] / inf["nnz"] if K + k2 else 0.0
                y = m.spmv_host(x)
                if y0 is None:
                    y0 = y
                same = bool(np.array_equal(y.view(np.uint8), y0.view(np.uint8)))
                v = np.array(acc[i])
                print("%-8s %s hub  table %8d B (%7d columns): LDS covers %.3f, both tiers <= %.3f of the entries   median %8.2f us  min %8.2f   "
                      "(%.0f GB/s algorithmic)  bits equal to the first layout's: %s" % (
                          case, dts, tables[i], k2, cov0, cov1, np.median(v), v.min(), inf["algorithmic_bytes"] / np.median(v) / 1e3, same), flush=True)
                m.close()
            v = np.array(acc[len(ms)])
            print("%-8s %s stream                                                                              median %8.2f us  min %8.2f" % (
                case, dts, np.median(v), v.min()), flush=True)
            m0.close()
            be.free(dx)
            be.free(dy)


if __name__ == "__main__":
    main()
