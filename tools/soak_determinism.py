import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import pkgload
pkg = pkgload.load_package()
be = pkg.Backend()
for cls, plans in (("C", [pkg.PLAN_AUTO, pkg.PLAN_TILED, pkg.PLAN_STREAM]), ("B", [pkg.PLAN_AUTO])):
    m = pkg.npb.NpbMatrix(cls)
    dm = be.matrix(m.a, m.rowstr, m.colidx, m.rows)
    x = np.random.default_rng(3).standard_normal(m.na + 2)
    for plan in plans:
        dm.set_plan(plan, 0)
        y0 = dm.spmv_host(x)
        bad = 0
        for i in range(150):
            y = dm.spmv_host(x)
            bad += not np.array_equal(y.view(np.uint64), y0.view(np.uint64))
        print(cls, pkg.abi.PLAN_NAMES[dm.info()["plan"]], "150 repeats, differing:", bad, flush=True)
    dm.close()
