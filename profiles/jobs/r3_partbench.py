import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import pkgload
pkg = pkgload.load_package()
be = pkg.Backend()
sb = pkg.callers.sb_generate(170)
rs, ci, a = (np.array(v) for v in sb.arrays())
cols = sb.rows
x = np.random.default_rng(1).standard_normal(cols + 2)
dx = be.malloc(x.nbytes); be.h2d(dx, x)
for parts in (2, 4, 8, 16):
    n = sb.rows // parts
    e = int(rs[n]) - 1
    rs_p, ci_p, a_p = rs[: n + 1].copy(), ci[:e].copy(), a[:e].copy()
    dm = be.matrix(a_p, rs_p, ci_p, n, cols_hint=cols)
    dy = be.malloc(n * 8 + 16)
    auto = pkg.abi.PLAN_NAMES[dm.info()["plan"]]
    line = "sb170 / %d: rows=%d nnz=%d auto=%s" % (parts, n, e, auto)
    for name, plan in (("stream", pkg.PLAN_STREAM), ("blocked", pkg.PLAN_BLOCKED)):
        dm.set_plan(plan, 0)
        dm.time_device(dx, dy, 5)
        line += " %s %.1f us" % (name, dm.time_device(dx, dy, 30) / 30 * 1e3)
    print(line, flush=True)
    dm.close(); be.free(dy)
be.release()
