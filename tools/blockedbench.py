#!/usr/bin/env python3
"""BLOCKED plan on SparseBench-like matrices of several sizes: bin sizes against STREAM.  GPU box only.
   python3 tools/blockedbench.py [size ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pkgload  # noqa: E402
pkg = pkgload.load_package()
be = pkg.Backend()
dtype = np.float32 if "--f32" in sys.argv else np.float64
sizes = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [170, 100]
for size in sizes:
    sb = pkg.callers.sb_generate(size)
    rs, ci, a = sb.arrays()
    a = a.astype(dtype)
    for extra in [int(v[7:]) for v in sys.argv[1:] if v.startswith("--plant")]:  # one row gets `extra` more entries (scattered)
        r0 = sb.rows // 3
        p0 = int(rs[r0 + 1]) - 1
        rng = np.random.default_rng(extra)
        ci = np.concatenate([ci[:p0], rng.integers(1, sb.rows + 1, extra).astype(np.int32), ci[p0:]])
        a = np.concatenate([a[:p0], rng.standard_normal(extra).astype(dtype), a[p0:]])
        rs = np.array(rs).copy()
        rs[r0 + 1:] += extra
    dm = be.matrix(a, rs, ci, sb.rows)
    inf = dm.info()
    x = np.random.default_rng(1).standard_normal(inf["cols"] + 2).astype(dtype)
    dx, dy = be.malloc(x.nbytes), be.malloc(sb.rows * 8 + 16)
    be.h2d(dx, x)
    dm.set_plan(pkg.PLAN_STREAM, 0)
    dm.time_device(dx, dy, 5)
    t_stream = dm.time_device(dx, dy, 20) / 20 * 1e3
    auto = pkg.abi.PLAN_NAMES[inf["plan"]]
    dm.close()
    line = "sb%d rows=%d nnz=%d x=%.1f MB auto=%s stream %.1f us;" % (size, sb.rows, inf["nnz"], inf["cols"] * 8 / 1e6, auto, t_stream)
    for bin_rows in [int(v[6:]) for v in sys.argv[1:] if v.startswith("--bins")] or (0,):
        be.set_tuning(18, bin_rows)
        dm = be.matrix(a, rs, ci, sb.rows)
        import time
        t0 = time.perf_counter()
        dm.set_plan(pkg.PLAN_BLOCKED, 0)
        t_build = time.perf_counter() - t0
        dm.time_device(dx, dy, 5)
        line += " blocked/%d %.1f us (built in %.0f ms)" % (bin_rows, dm.time_device(dx, dy, 20) / 20 * 1e3, t_build * 1e3)
        dm.close()
    be.set_tuning(18, 0)
    print(line, flush=True)
    be.free(dx), be.free(dy)
    del rs, ci, a
    pkg.callers.free(sb)
be.release()  # before the interpreter exits: under rocprofv3 the library's exit handler would run after the profiler has shut down
