#!/usr/bin/env python3
"""Debug aid: the STRIPED layout built by the host threads against the one built on the GPU, array by array (debug
library).  GPU box only.    python tools/striped_builder_diff.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import matgen  # noqa: E402
import pkgload  # noqa: E402

pkg = pkgload.load_package()
be = pkg.Backend(pkg.lib_path("libhip-spmv-debug.so"))
be.lib.spmv_hip_debug_striped_array.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
be.lib.spmv_hip_debug_striped_array.restype = ctypes.c_longlong
NAMES = ["tptr", "rrow", "wgd", "col", "val"]
DT = [np.int32, np.int32, np.int32, np.uint16, np.float64]


def arrays(m):
    out = []
    for k in range(5):
        n = be.lib.spmv_hip_debug_striped_array(m.h, k, None, 0)
        buf = np.zeros(n, np.uint8)
        assert be.lib.spmv_hip_debug_striped_array(m.h, k, buf.ctypes.data, n) == n
        out.append(buf.view(DT[k]))
    return out


cases = {"wide_40_stripes": matgen.random_csr(35, 3000, 600000, 90, np.float64, spread=0.3),
         "npb_like": matgen.npb_like(36, 20000, 200, np.float64)}
for name, (rs, ci, a) in cases.items():
    rows = rs.size - 1
    got = []
    for builder in (1, 2):
        be.set_tuning(17, builder)
        m = be.matrix(a, rs, ci, rows)
        m.set_plan(pkg.PLAN_STRIPED, 0)
        got.append(arrays(m))
        x = np.random.default_rng(1).standard_normal(int(ci.max()))
        y = m.spmv_host(x)
        print(name, "builder", builder, "info", m.info()["n_blocks"], "|y|max", float(np.abs(y).max()))
        m.close()
    for k in range(5):
        h, d = got[0][k], got[1][k]
        if h.size != d.size:
            print("  %s: sizes differ %d vs %d" % (NAMES[k], h.size, d.size))
            continue
        neq = np.nonzero(h != d)[0] if k != 4 else np.nonzero(h.view(np.uint64) != d.view(np.uint64))[0]
        print("  %s: %d entries, %d differ%s" % (NAMES[k], h.size, neq.size, "" if not neq.size else
              "; first at %d: host %r device %r; host[..8] %r device[..8] %r" % (neq[0], h[neq[0]], d[neq[0]], h[neq[0]:neq[0] + 8], d[neq[0]:neq[0] + 8])))
