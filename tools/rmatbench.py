#!/usr/bin/env python3
"""R-MAT graphs (the C generator of the bfs / pagerank counterparts): AUTO, STREAM and BLOCKED by name.  GPU box only.
   python3 tools/rmatbench.py [scale ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pkgload  # noqa: E402
pkg = pkgload.load_package()
be = pkg.Backend()
for sc in [int(v) for v in sys.argv[1:]] or [20, 21]:
    g = pkg.callers.graph_rmat(sc, 16)
    rs, ci, a = g.arrays()
    dm = be.matrix(a, rs, ci, g.rows)
    inf = dm.info()
    x = np.random.default_rng(1).standard_normal(inf["cols"] + 2)
    dx, dy = be.malloc(x.nbytes), be.malloc(g.rows * 8 + 16)
    be.h2d(dx, x)

    def t():
        dm.time_device(dx, dy, 5)
        return dm.time_device(dx, dy, 20) / 20 * 1e3
    line = "rmat%d rows=%d nnz=%d max_row=%d x=%.1f MB auto=%s %.1f us;" % (sc, g.rows, inf["nnz"], inf["max_row_len"], inf["cols"] * 8 / 1e6,
                                                                 pkg.abi.PLAN_NAMES[inf["plan"]], t())
    dm.set_plan(pkg.PLAN_STREAM, 0)
    line += " stream %.1f us;" % t()
    y_ref = be.d2h(np.empty(g.rows), dy)
    try:
        dm.set_plan(pkg.PLAN_BLOCKED, 0)
        i2 = dm.info()
        line += " blocked %.1f us (%d launches)" % (t(), i2["n_launches"])
        y = be.d2h(np.empty(g.rows), dy)
        line += " max|dy|/max|y| %.1e" % (np.abs(y - y_ref).max() / np.abs(y_ref).max())
    except RuntimeError as e:
        line += " blocked: %s" % e
    print(line, flush=True)
    dm.close()
    be.free(dx), be.free(dy)
    del rs, ci, a
    pkg.callers.free(g)
be.release()
