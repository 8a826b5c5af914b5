#!/usr/bin/env python3
"""Is BLOCKED's result a function of the layout?  The same product many times over, plain adds and run sums, both precisions,
matrices with many same-row collisions inside one LDS instruction; every repetition compared bit for bit.  GPU box only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pkgload  # noqa: E402
pkg = pkgload.load_package()
be = pkg.Backend()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cases = []
sb = pkg.callers.sb_generate(100)
cases.append(("sparsebench 100^3 (plain adds)",) + tuple(np.array(v) for v in sb.arrays()))
pkg.callers.free(sb)
g = pkg.callers.graph_rmat(19, 16)
cases.append(("R-MAT 2^19 (run sums, cut bins)",) + tuple(np.array(v) for v in g.arrays()))
pkg.callers.free(g)
for name, rs, ci, a in cases:
    for dtype in (np.float64, np.float32):
        rows = rs.size - 1
        x = np.random.default_rng(3).standard_normal(int(ci.max())).astype(dtype)
        dm = be.matrix(a.astype(dtype), rs, ci, rows)
        dm.set_plan(pkg.PLAN_BLOCKED, 0)
        y0 = dm.spmv_host(x)
        view = np.uint64 if dtype == np.float64 else np.uint32
        bad = sum(not np.array_equal(dm.spmv_host(x).view(view), y0.view(view)) for _ in range(reps))
        print("%s %s: %d repetitions, %d differ from the first (launches %d)" % (name, np.dtype(dtype).name, reps, bad, dm.info()["n_launches"]), flush=True)
        dm.close()
be.release()
