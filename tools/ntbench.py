#!/usr/bin/env python3
"""k_striped's packed stream loaded normally or non-temporally (tuning key 20), by stream size: where does the Infinity
Cache stop helping a repeated product?  Row blocks of NPB class C of growing size, classes B, C, D/8, D/2; interleaved
rounds on one matrix object.  GPU box only.     python tools/ntbench.py [--cases ...]"""
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
    ap.add_argument("--cases", default="B,C:60000,C:82000,C:95000,C:106000,C:123000,C,C/f32,D/8,D/2")
    ap.add_argument("--rounds", type=int, default=10)
    args = ap.parse_args()
    pkg = pkgload.load_package()
    be = pkg.Backend()
    for case in args.cases.split(","):
        dtype = np.float32 if case.endswith("/f32") else np.float64
        name = case[:-4] if case.endswith("/f32") else case
        if ":" in name:
            cls, k = name.split(":")
            keep = pkg.npb.NpbMatrix(cls, rows=(0, int(k)))
            rs, ci, a = keep.rowstr, keep.colidx, keep.a.astype(dtype)
            hint = keep.na
        else:
            rs, ci, a, keep = build_case(pkg, name, dtype)
            hint = keep.na if name.startswith("D/") else 0
        rows = rs.size - 1
        dm = be.matrix(a, rs, ci, rows, cols_hint=hint)
        dm.set_plan(pkg.PLAN_STRIPED, 0)
        inf = dm.info()
        x = np.random.default_rng(1).standard_normal(inf["cols"] + 2).astype(dtype)
        dx, dy = be.malloc(x.nbytes), be.malloc(rows * dtype().itemsize + 16)
        be.h2d(dx, x)
        acc = {1: [], 2: []}
        dm.time_device(dx, dy, 30)
        for _ in range(args.rounds):
            for mode in (2, 1):
                be.set_tuning(20, mode)
                dm.time_device(dx, dy, 3)
                acc[mode].append(dm.time_device(dx, dy, 20) / 20 * 1e3)
        be.set_tuning(20, 0)
        stream_mb = inf["nnz"] * (dtype().itemsize + 2) / 1e6  # without the pads (a few per cent)
        print("%-10s stream ~%6.0f MB   normal %.2f us   non-temporal %.2f us   (%+.1f %%)   alg frac normal %.3f / nt %.3f" % (
            case, stream_mb, np.median(acc[2]), np.median(acc[1]), (np.median(acc[1]) / np.median(acc[2]) - 1) * 100,
            inf["algorithmic_bytes"] / np.median(acc[2]) / 8e6, inf["algorithmic_bytes"] / np.median(acc[1]) / 8e6), flush=True)
        dm.close()
        be.free(dx), be.free(dy)
        del keep


if __name__ == "__main__":
    main()
