#!/usr/bin/env python3
"""Where the end of a k_striped launch comes from: per-workgroup clock stamps (debug library) over many launches, set
against what each workgroup holds (stripes, entries, rows) and where it runs (XCD).  GPU box only.

    python tools/wgtimes.py --case D/8 --launches 40 [--shape 1]
"""
import argparse
import ctypes
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
    ap.add_argument("--case", default="D/8")
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--shape", type=int, default=0)
    ap.add_argument("--groups", type=int, default=0)
    ap.add_argument("--dump", action="store_true")
    args = ap.parse_args()
    pkg = pkgload.load_package()
    be = pkg.Backend(pkg.lib_path("libhip-spmv-debug.so"))
    rs, ci, a, keep = build_case(pkg, args.case, np.float64)
    rows = rs.size - 1
    be.set_tuning(15, args.shape)
    be.set_tuning(11, args.groups)
    dm = be.matrix(a, rs, ci, rows, cols_hint=(keep.na if args.case.startswith("D/") else 0))
    dm.set_plan(pkg.PLAN_STRIPED, 0)
    inf = dm.info()
    x = np.random.default_rng(1).standard_normal(inf["cols"] + 2)
    dx = be.malloc(x.nbytes)
    dy = be.malloc(rows * 8 + 16)
    be.h2d(dx, x)
    nwg = inf["n_blocks"]
    lay = (ctypes.c_longlong * (6 * nwg))()
    be.lib.spmv_hip_debug_striped_wgs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert be.lib.spmv_hip_debug_striped_wgs(dm.h, lay, nwg) == nwg
    lay = np.array(lay[:], dtype=np.int64).reshape(nwg, 6)
    ns, ent, longest, wrows = lay[:, 1] - lay[:, 0], lay[:, 3], lay[:, 4], lay[:, 5]
    dm.time_device(dx, dy, 60)
    ms = dm.time_device(dx, dy, 40) / 40
    print("# %s: %d workgroups, %.2f us per product (plain launches)" % (args.case, nwg, ms * 1e3))
    buf = (ctypes.c_ulonglong * (4 * nwg))()
    be.lib.spmv_hip_debug_stamps.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    ends, stag = [], []
    for _ in range(args.launches):
        be.set_tuning(6, 64)
        dm.time_device(dx, dy, 1)
        be.set_tuning(6, 0)
        assert be.lib.spmv_hip_debug_stamps(0, buf, 4 * nwg) == 0
        st = np.array(buf[:], dtype=np.float64).reshape(nwg, 4) * 0.01
        t0 = st[:, 0].min()
        ends.append(st[:, 3] - t0)
        stag.append(st[:, 1] - st[:, 0])
    ends = np.array(ends).mean(axis=0)
    per_xcd = max(1, (nwg + 7) // 8)
    xcd = np.arange(nwg) // per_xcd  # tuning 0 = 1: index // per_xcd is the XCD
    print("# mean end of a workgroup after the launch's first entry: min %.1f  median %.1f  max %.1f us; staging %.2f us" % (
        ends.min(), np.median(ends), ends.max(), np.mean(stag)))
    for k in sorted(set(ns.tolist())):
        sel = ns == k
        print("#   %d stripes: %3d workgroups, entries %.0f +- %.0f (longest wave %.0f), rows %.0f, end %.1f +- %.1f us" % (
            k, sel.sum(), ent[sel].mean(), ent[sel].std(), longest[sel].mean(), wrows[sel].mean(), ends[sel].mean(), ends[sel].std()))
    print("#   by XCD: " + " ".join("%d:%.1f" % (k, ends[xcd == k].mean()) for k in range(8)))
    # least squares: end = a * entries + b * stripes + c_xcd
    A = np.zeros((nwg, 10))
    A[:, 0] = ent / 1e6
    A[:, 1] = ns
    A[np.arange(nwg), 2 + xcd] = 1.0
    coef, *_ = np.linalg.lstsq(A, ends, rcond=None)
    res = ends - A @ coef
    print("# fit: end = %.2f us per M entries + %.2f us per stripe + XCD offsets [%s]; residual std %.2f us" % (
        coef[0], coef[1], " ".join("%.1f" % c for c in coef[2:]), res.std()))
    if args.dump:
        for q in range(nwg):
            print("%d g=%d ns=%d ent=%d longest=%d rows=%d xcd=%d end=%.1f" % (q, lay[q, 2], ns[q], ent[q], longest[q], wrows[q], xcd[q], ends[q]))


if __name__ == "__main__":
    main()
