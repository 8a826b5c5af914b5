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
PERMUTE = "--permute" in sys.argv  # rows and columns scrambled in groups of 8 (64-byte lines of y and x): what a BLOCKED layout
                                   # over a pseudo-random permutation would see (feasibility: the x / y permutation itself is not timed)


def scramble(n, mult):
    """a bijection of [0, n8 * 8) that keeps groups of 8 together: line l -> (l * mult) mod n8, mult coprime to n8"""
    n8 = (n + 7) // 8
    while np.gcd(mult, n8) != 1:
        mult += 1
    idx = np.arange(n8 * 8, dtype=np.int64)
    return ((idx // 8) * mult % n8) * 8 + idx % 8


for sc in [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [20, 21]:
    g = pkg.callers.graph_rmat(sc, 16)
    rs, ci, a = (np.array(v) for v in g.arrays())
    if PERMUTE:
        rows = g.rows
        pr = scramble(rows, 0x9E3779B1 % (1 << 30))  # new index of every old row (rows is a power of two here: in range)
        pc = scramble(int(ci.max()), 0x85EBCA6B % (1 << 30))
        rowid = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rs))
        nr, nc = pr[rowid], pc[ci.astype(np.int64) - 1]
        order = np.argsort(nr, kind="stable")
        lens = np.bincount(nr, minlength=rows)[:rows]
        assert lens.sum() == ci.size
        rs = np.concatenate(([1], 1 + np.cumsum(lens))).astype(np.int32)
        ci = (nc[order] + 1).astype(np.int32)
        a = a[order]
    dm = be.matrix(a, rs, ci, g.rows)
    inf = dm.info()
    x = np.random.default_rng(1).standard_normal(inf["cols"] + 2)
    dx, dy = be.malloc(x.nbytes), be.malloc(g.rows * 8 + 16)
    be.h2d(dx, x)

    def t():
        dm.time_device(dx, dy, 5)
        return dm.time_device(dx, dy, 20) / 20 * 1e3
    line = ("permuted " if PERMUTE else "") + "rmat%d rows=%d nnz=%d max_row=%d x=%.1f MB auto=%s %.1f us;" % (sc, g.rows, inf["nnz"], inf["max_row_len"], inf["cols"] * 8 / 1e6,
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
