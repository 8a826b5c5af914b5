#!/usr/bin/env python3
"""What would BLOCKED gain from a leaner byte budget?  Timing-only probes of the debug library (tuning key 26; results are
wrong): 1 = phase 1 without the destination stream (-1 B per entry read; the products land where they were read, i.e. also
perfectly sequential writes), 2 = phase 2 without the row stream (-2 B per entry read), 3 = both.  Interleaved with the real
kernels.  GPU box only.   python3 tools/blockedabl.py [size ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pkgload  # noqa: E402
pkg = pkgload.load_package()
be = pkg.Backend(pkg.lib_path("libhip-spmv-debug.so"))
for size in [int(v) for v in sys.argv[1:]] or [170, 128]:
    sb = pkg.callers.sb_generate(size)
    rs, ci, a = sb.arrays()
    dm = be.matrix(a, rs, ci, sb.rows)
    dm.set_plan(pkg.PLAN_BLOCKED, 0)
    inf = dm.info()
    x = np.random.default_rng(1).standard_normal(inf["cols"] + 2)
    dx, dy = be.malloc(x.nbytes), be.malloc(sb.rows * 8 + 16)
    be.h2d(dx, x)
    acc = {m: [] for m in (0, 1, 2, 3)}
    for _ in range(6):
        for m in acc:
            be.set_tuning(26, m)
            dm.time_device(dx, dy, 3)
            acc[m].append(dm.time_device(dx, dy, 20) / 20 * 1e3)
    be.set_tuning(26, 0)
    print("# SparseBench %d^3: rows=%d nnz=%d launches=%d, 6 interleaved rounds x 20 products" % (size, sb.rows, inf["nnz"], inf["n_launches"]), flush=True)
    for m, name in ((0, "the plan as shipped (30 B per entry moved)"), (1, "phase 1 without dest4, sequential product writes"),
                    (2, "phase 2 without lrow"), (3, "both (27 B per entry)")):
        v = np.array(acc[m])
        print("  %-52s median %7.2f us  min %7.2f" % (name, np.median(v), v.min()), flush=True)
    dm.close(); be.free(dx); be.free(dy)
    del rs, ci, a
    pkg.callers.free(sb)
be.release()
