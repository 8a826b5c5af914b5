#!/usr/bin/env python3
"""What would a hot / cold column split buy on power-law graphs?  A = A_hot + A_cold by column popularity: the K most
popular columns compacted into their own matrix (x_hot = x[hot], K columns: one LDS stripe for STRIPED), the rest left to
the CSR kernel.  Existing plans only: two matrix objects timed one after the other against the whole matrix.  GPU box only.
   python3 tools/hotcoldbench.py [scale ...] [--k16384]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pkgload  # noqa: E402
pkg = pkgload.load_package()
be = pkg.Backend()
K = [int(v[3:]) for v in sys.argv[1:] if v.startswith("--k")] or [16384]
for sc in [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [20]:
    g = pkg.callers.graph_rmat(sc, 16)
    rs, ci, a = (np.array(v) for v in g.arrays())
    rows, cols = g.rows, int(ci.max())
    pop = np.bincount(ci - 1, minlength=cols)
    order = np.argsort(-pop, kind="stable")
    x = np.random.default_rng(1).standard_normal(cols + 2)

    def timed(a_, rs_, ci_, ncols, xv, plan=None):
        dm = be.matrix(a_, rs_, ci_, rows, cols_hint=ncols)
        if plan is not None:
            dm.set_plan(plan, 0)
        dx, dy = be.malloc(xv.nbytes + 64), be.malloc(rows * 8 + 16)
        be.h2d(dx, xv)
        dm.time_device(dx, dy, 5)
        t = dm.time_device(dx, dy, 20) / 20 * 1e3
        name = pkg.abi.PLAN_NAMES[dm.info()["plan"]]
        y = be.d2h(np.empty(rows), dy)
        dm.close(), be.free(dx), be.free(dy)
        return t, name, y
    t_full, p_full, y_full = timed(a, rs, ci, cols, x)
    line = "rmat%d nnz=%d: whole %s %.1f us;" % (sc, ci.size, p_full, t_full)
    rowid = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rs))
    for k in K:
        rank = np.empty(cols, np.int64)
        rank[order] = np.arange(cols)
        hot = rank[ci - 1] < k
        parts = []
        for sel, remap in ((hot, True), (~hot, False)):
            lens = np.bincount(rowid[sel], minlength=rows)
            rs_ = np.concatenate(([1], 1 + np.cumsum(lens))).astype(np.int32)
            ci_ = (rank[ci[sel] - 1] + 1).astype(np.int32) if remap else ci[sel]
            parts.append((a[sel], rs_, ci_))
        xh = np.concatenate((x[order[:k]], np.zeros(2)))
        t_hot, p_hot, y_hot = timed(parts[0][0], parts[0][1], parts[0][2], k, xh)
        t_cold, p_cold, y_cold = timed(parts[1][0], parts[1][1], parts[1][2], cols, x)
        err = np.abs(y_hot + y_cold - y_full).max() / np.abs(y_full).max()
        line += " K=%d hot %.0f %% of the entries: %s %.1f us + cold %s %.1f us = %.1f (max|dy|/max|y| %.1e);" % (
            k, 100.0 * hot.mean(), p_hot, t_hot, p_cold, t_cold, t_hot + t_cold, err)
    print(line, flush=True)
    pkg.callers.free(g)
be.release()
