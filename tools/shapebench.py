#!/usr/bin/env python3
"""STRIPED geometry A/B: builds one layout per configuration (kernel shape, stripe groups, workgroups aimed for) of the
same matrix, checks every one against the CSR row-segment kernel, and times them in turn (interleaved rounds: boxes
drift by a few per cent within a run).  GPU box only.

    python tools/shapebench.py --cases D/8,D/4,C --configs 1:0:0,2:0:0,2:0:768 --rounds 8

A configuration is shape:groups:workgroups[:xcd_bias] (0 = the builder's choice; xcd_bias in per mille); shape 1 = 16 waves x 128 KiB of x (one workgroup
per CU), 2 = 8 waves x 64 KiB (two per CU).
"""
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
    ap.add_argument("--cases", default="D/8,C")
    ap.add_argument("--configs", default="1:0:0,2:0:0")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--allocs", type=int, default=1, help="matrix objects per configuration (two allocations of one layout differ by up to 4 %)")
    args = ap.parse_args()
    pkg = pkgload.load_package()
    be = pkg.Backend()
    dtype = np.float64 if args.dtype == "f64" else np.float32
    cfgs = [tuple(int(v) for v in c.split(":")) for c in args.configs.split(",")]
    for case in args.cases.split(","):
        rs, ci, a, keep = build_case(pkg, case, dtype)
        rows = rs.size - 1
        hint = keep.na if case.startswith("D/") else 0
        ref = be.matrix(a, rs, ci, rows, cols_hint=hint)
        ref.set_plan(pkg.PLAN_STREAM, 0)
        inf = ref.info()
        x = np.random.default_rng(1).standard_normal(inf["cols"] + 2).astype(dtype)
        dx = be.malloc(x.nbytes)
        dy = be.malloc(rows * dtype().itemsize + 16)
        be.h2d(dx, x)
        ref.spmv_device(dx, dy)
        y_ref = be.d2h(np.empty(rows, dtype), dy)
        ref.close()
        print("# %s: rows=%d cols=%d nnz=%d alg_bytes=%d" % (case, rows, inf["cols"], inf["nnz"], inf["algorithmic_bytes"]), flush=True)
        mats = []
        for cfg in [c for c in cfgs for _ in range(args.allocs)]:
            shape, ng, wgs = cfg[:3]
            bias = cfg[3] if len(cfg) > 3 else 0
            be.set_tuning(15, shape)
            be.set_tuning(11, ng)
            be.set_tuning(9, wgs)
            be.set_tuning(19, bias)  # per mille of weight moved from even-XCD workgroups to odd-XCD ones
            dm = be.matrix(a, rs, ci, rows, cols_hint=hint)
            try:
                dm.set_plan(pkg.PLAN_STRIPED, 0)
            except RuntimeError as e:
                print("%-6s %s unavailable: %s" % (case, cfg, e), flush=True)
                dm.close()
                continue
            dm.spmv_device(dx, dy)
            y = be.d2h(np.empty(rows, dtype), dy)
            err = float(np.max(np.abs(y - y_ref)) / max(1e-300, float(np.max(np.abs(y_ref)))))
            dm.time_device(dx, dy, 30)
            mats.append(((shape, ng, wgs, bias, len(mats)), dm, dm.info(), err))
        be.set_tuning(15, 0), be.set_tuning(11, 0), be.set_tuning(9, 0), be.set_tuning(19, 0)
        acc = {k: [] for k, *_ in mats}
        for _ in range(args.rounds):
            for k, dm, _, _ in mats:
                dm.time_device(dx, dy, 2)
                acc[k].append(dm.time_device(dx, dy, args.reps) / args.reps * 1e3)
        if args.allocs > 1:  # per configuration over its allocations
            for cfg in cfgs:
                key = tuple(cfg[:3]) + ((cfg[3] if len(cfg) > 3 else 0),)
                vs = [np.median(acc[k]) for k, *_ in mats if k[:4] == key]
                if vs:
                    print("%-6s shape=%d groups=%d wgs=%d xcd_bias=%d: medians of %d allocations %s  -> mean %.2f us" % (
                        (case,) + key + (len(vs), " ".join("%.2f" % v for v in vs), float(np.mean(vs)))), flush=True)
        for k, dm, i2, err in mats:
            v = np.array(acc[k])
            print("%-6s shape=%d groups=%d wgs=%d xcd_bias=%d -> workgroups=%d launches=%d  mean %.2f us  median %.2f  min %.2f  std %.2f   %.1f GB/s(alg) frac %.3f  max|dy|/max|y| %.1e" % (
                case, k[0], k[1], k[2], k[3], i2["n_blocks"], i2["n_launches"], v.mean(), np.median(v), v.min(), v.std(),
                i2["algorithmic_bytes"] / np.median(v) / 1e3, i2["algorithmic_bytes"] / np.median(v) / 1e3 / 8000.0, err), flush=True)
            dm.close()
        be.free(dx)
        be.free(dy)
        del keep


if __name__ == "__main__":
    main()
