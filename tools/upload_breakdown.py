"""tools/upload_breakdown.py -- where the one-time cost of a STRIPED matrix goes: creating the device matrix (CSR upload,
statistics, column range, re-layout on the host, upload of the layout) against its parts timed alone."""
import ctypes
import sys
import time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pkgload, conftest, test_plan
pkg = pkgload.load_package()
be = pkg.Backend()
lib = ctypes.CDLL(conftest.build_planlib())
for cls in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["C"]):
    m = pkg.npb.NpbMatrix(cls)
    be.malloc(16)
    for rep in range(2):
        t0 = time.perf_counter()
        dm = be.matrix(m.a, m.rowstr, m.colidx, m.rows, cols_hint=m.na)
        t1 = time.perf_counter()
        plan = dm.info()["plan"]
        dm.close()
        print("%s create (plan %s): %.3f s" % (cls, pkg.abi.PLAN_NAMES[plan], t1 - t0), flush=True)
    t0 = time.perf_counter()
    rc, t = test_plan.striped_build(lib, m.rowstr, m.colidx, m.a, xbytes=131072, wpb=16, wgs=256, wave_rows=1024, groups=0)
    t1 = time.perf_counter()
    lib.spmv_build_threads.restype = ctypes.c_int
    print("%s host re-layout alone (%d threads): %.3f s  (%d packed entries)" % (cls, lib.spmv_build_threads(), t1 - t0, t.n_packed))
    n_packed = t.n_packed
    lib.spmv_striped_free.argtypes = [ctypes.POINTER(test_plan.StripedHost)]
    lib.spmv_striped_free(ctypes.byref(t))
    d = be.malloc(m.a.nbytes)
    for name, arr in (("a (pageable)", m.a), ("colidx (pageable)", m.colidx)):
        t0 = time.perf_counter(); be.h2d(d, arr); t1 = time.perf_counter()
        print("%s H2D %s %.0f MB: %.3f s = %.1f GB/s" % (cls, name, arr.nbytes / 1e6, t1 - t0, arr.nbytes / (t1 - t0) / 1e9))
    be.free(d)
