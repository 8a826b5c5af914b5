#!/usr/bin/env python3
"""tools/ref_callers_compare.py -- the reference's own programs (compiled from its sources into oracle/_ref by
oracle/Makefile) on the reference's CPU backend and on libhip-spmv.so, same input, same machine.

  bfs      bfs/bfs.cc: 16 BFS from random sources on a graph read from stdin; prints its own elapsed seconds.
           Input: a seeded R-MAT graph written in the program's "rows cols nnz / x y" format.
  parboil  src/cpu/main.c: 50 products; prints its own Compute time.  Inputs: the bundled datasets.

SPMV_HIP_DISPATCH is taken from the environment (default auto: small products run on the library's host loop and the
HIP context is created only for a matrix that is worth it, csrc/spmv_harness.c dispatch_host).
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import matgen  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def write_pairs(path, scale, edge_factor):
    rs, ci, a = matgen.rmat_graph(20, scale, edge_factor, np.float64, values="ones")
    rows = rs.size - 1
    r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rs))
    c = ci.astype(np.int64) - 1
    # library.cc:56-80 shifts rows by one and adds one to the column: write pairs that survive it
    with open(path, "w") as f:
        f.write("%d %d %d\n" % (rows, rows, r.size))
        np.savetxt(f, np.stack([r, c], axis=1), fmt="%d")
    return rows, r.size


def main():
    scales = [int(s) for s in sys.argv[1:]] or [16, 18]
    with tempfile.TemporaryDirectory() as tmp:
        for sc in scales:
            path = os.path.join(tmp, "rmat%d.pairs" % sc)
            rows, nnz = write_pairs(path, sc, 16)
            line = "bfs R-MAT 2^%d (%d vertices, %d edges):" % (sc, rows, nnz)
            for name in ("bfs_native", "bfs_hip"):
                exe = os.path.join(REF, name)
                if not os.path.exists(exe):
                    line += "  %s missing" % name
                    continue
                best = None
                for _ in range(3):
                    with open(path) as f:
                        r = subprocess.run([exe], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                           timeout=1500)
                    if r.returncode != 0:
                        line += "  %s failed (%s)" % (name, r.stderr[-200:])
                        break
                    t = float(r.stdout.strip().splitlines()[-1])
                    best = t if best is None else min(best, t)
                if best is not None:
                    line += "  %s %.3f s" % (name, best)
            print(line, flush=True)
        gold = os.path.join(ROOT, "tests", "golden")
        for ds, mtx in (("parboil_small", "1138_bus.mtx"), ("parboil_medium", "bcsstk18.mtx")):
            line = "parboil %s:" % ds
            for name in ("parboil_spmv_oracle", "parboil_spmv_hip"):
                exe = os.path.join(REF, name)
                if not os.path.exists(exe):
                    continue
                r = subprocess.run([exe, "-i", os.path.join(gold, ds, mtx) + "," + os.path.join(gold, ds, "vector.bin"),
                                    "-o", os.path.join(tmp, "o.bin")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   text=True, cwd=tmp, timeout=300)
                m = re.search(r"Compute\s*:\s*([0-9.]+)", r.stdout)
                line += "  %s Compute %s s" % (name, m.group(1) if m else "?")
            print(line, flush=True)


if __name__ == "__main__":
    main()
