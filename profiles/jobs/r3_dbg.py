import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import pkgload
pkg = pkgload.load_package()
be = pkg.Backend()
sb = pkg.callers.sb_generate(130)
rs, ci, a = (np.array(v) for v in sb.arrays())
m = be.matrix(a, rs, ci, sb.rows)
inf = m.info()
print({k: inf[k] for k in ("plan", "plan_param", "cols", "rows", "nnz", "max_row_len")}, flush=True)
try:
    print("last error:", be.last_error())
except Exception as e:
    print("no last_error", e)
m.set_plan(pkg.PLAN_AUTO, 0)
print(m.info()["plan"], flush=True)
m.close()
be.release()
