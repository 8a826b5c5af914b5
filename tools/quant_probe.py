"""tools/quant_probe.py -- what do 25 / 26 row blocks per stripe group cost?  NPB-like matrices (240 nonzeros per row,
uniform sorted columns) of 131056 columns (8 stripes of 16382: 8 groups x 32 workgroups, no rounding) and of 150000
(10 stripes: groups of 25 and 26 workgroups, like class C), same rows x nonzeros per row; ns per nonzero, interleaved."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pkgload, matgen
pkg = pkgload.load_package()
be = pkg.Backend()
ms = {}
for n in (131056, 150000, 163820, 180202):
    rs, ci, a = matgen.npb_like(5, n, 240)
    dm = be.matrix(a, rs, ci, n)
    dm.set_plan(pkg.PLAN_STRIPED, 0)
    x = np.random.default_rng(1).standard_normal(n)
    dx = be.malloc(x.nbytes); dy = be.malloc(n * 8); be.h2d(dx, x)
    dm.time_device(dx, dy, 200)
    ms[n] = (dm, dx, dy, a.size, dm.info())
for rnd in range(6):
    for n, (dm, dx, dy, nnz, inf) in ms.items():
        t = dm.time_device(dx, dy, 100) / 100
        print("round %d  n=%6d nnz=%9d  blocks=%d  %.2f us  %.4f ns/nnz" % (rnd, n, nnz, inf["n_blocks"], t * 1e3, t * 1e6 / nnz), flush=True)
