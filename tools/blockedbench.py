#!/usr/bin/env python3
"""BLOCKED plan on SparseBench-like matrices: phase-1 grid sweep against STREAM.  GPU box only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pkgload  # noqa: E402
pkg = pkgload.load_package()
be = pkg.Backend()
for size in (170, 100):
    sb = pkg.callers.sb_generate(size)
    rs, ci, a = sb.arrays()
    dm = be.matrix(a, rs, ci, sb.rows)
    inf = dm.info()
    x = np.random.default_rng(1).standard_normal(inf["cols"] + 2)
    dx, dy = be.malloc(x.nbytes), be.malloc(sb.rows * 8 + 16)
    be.h2d(dx, x)
    dm.set_plan(pkg.PLAN_STREAM, 0)
    dm.time_device(dx, dy, 5)
    print("sb%d stream %.1f us" % (size, dm.time_device(dx, dy, 20) / 20 * 1e3), flush=True)
    dm.set_plan(pkg.PLAN_BLOCKED, 0)
    for wgs in (64, 128, 256, 512, 1024):
        be.set_tuning(18, wgs)
        dm.time_device(dx, dy, 5)
        print("sb%d blocked wgs/xcd=%d %.1f us" % (size, wgs, dm.time_device(dx, dy, 20) / 20 * 1e3), flush=True)
    be.set_tuning(18, 256)
    dm.close()
    be.free(dx), be.free(dy)
    del rs, ci, a
    pkg.callers.free(sb)
