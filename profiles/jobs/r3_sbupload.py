import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import pkgload
pkg = pkgload.load_package()
be = pkg.Backend()
sb = pkg.callers.sb_generate(170)
rs, ci, a = (np.array(v) for v in sb.arrays())
be.malloc(16)
for rep in range(3):
    t0 = time.perf_counter(); dm = be.matrix(a, rs, ci, sb.rows); t1 = time.perf_counter()
    print("create (plan %s): %.1f ms" % (pkg.abi.PLAN_NAMES[dm.info()["plan"]], (t1 - t0) * 1e3), flush=True)
    t0 = time.perf_counter(); dm.refresh_values(a, ci); t1 = time.perf_counter()
    print("  refresh_values (193 MB of values H2D + the layout again): %.1f ms" % ((t1 - t0) * 1e3))
    dm.set_plan(pkg.PLAN_STREAM, 0)
    t0 = time.perf_counter(); dm.refresh_values(a, ci); t1 = time.perf_counter()
    print("  refresh_values under STREAM (values H2D, the stale layout dropped): %.1f ms" % ((t1 - t0) * 1e3))
    t0 = time.perf_counter(); dm.set_plan(pkg.PLAN_BLOCKED, 0); be.synchronize(0); t1 = time.perf_counter()
    print("  BLOCKED layout alone: %.1f ms" % ((t1 - t0) * 1e3))
    dm.close()
d = be.malloc(a.nbytes)
for name, arr in (("a", a), ("colidx", ci), ("rowstr", rs)):
    t0 = time.perf_counter(); be.h2d(d, arr); t1 = time.perf_counter()
    print("H2D %s %.0f MB pageable: %.1f ms = %.1f GB/s" % (name, arr.nbytes / 1e6, (t1 - t0) * 1e3, arr.nbytes / (t1 - t0) / 1e9))
be.free(d)
be.release()
