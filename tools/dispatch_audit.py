#!/usr/bin/env python3
"""tools/dispatch_audit.py -- where is AUTO not the fastest plan?  A grid of synthetic shapes (uniformly
scattered columns, NPB-like row-length spread) timed under every candidate plan with HIP events;
prints the shapes where the automatic choice is more than 10 % behind the best one."""
import sys
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import matgen  # noqa: E402
import pkgload  # noqa: E402


def main():
    pkg = pkgload.load_package()
    be = pkg.Backend()
    cand = [("striped", pkg.PLAN_STRIPED, 0), ("stream", pkg.PLAN_STREAM, 0), ("vec16", pkg.PLAN_VECTOR, 16),
            ("vec64", pkg.PLAN_VECTOR, 64), ("vec4", pkg.PLAN_VECTOR, 4)]
    print("%8s %6s %9s | %-8s %8s | best %-8s %8s | %s" % ("rows", "mean", "nnz", "auto", "us", "plan", "us", "all"))
    worst = []
    for rows in (2000, 8000, 32000, 128000, 512000, 2000000):
        for mean in (4, 16, 64, 256):
            if rows * mean > 40_000_000 or rows * mean < 20000:
                continue
            for cols_f in (1.0, 8.0):
                cols = int(rows * cols_f)
                if cols_f != 1.0 and (rows > 128000 or mean < 16):
                    continue
                rs, ci, a = matgen.npb_like(7, rows, mean)
                if cols_f != 1.0:  # spread the same pattern over more columns
                    ci = ((ci.astype(np.int64) - 1) * 8 + 1).astype(np.int32)
                dm = be.matrix(a, rs, ci, rows)
                inf = dm.info()
                x = np.random.default_rng(1).standard_normal(inf["cols"] + 2)
                dx = be.malloc(x.nbytes)
                dy = be.malloc(rows * 8 + 16)
                be.h2d(dx, x)

                def t():
                    dm.time_device(dx, dy, 3)
                    return dm.time_device(dx, dy, 40) / 40 * 1e3
                dm.set_plan(pkg.PLAN_AUTO, 0)
                auto_plan = pkg.abi.PLAN_NAMES[dm.info()["plan"]]
                if auto_plan == "vector":
                    auto_plan += str(dm.info()["plan_param"])
                auto_us = t()
                res = {}
                for name, plan, par in cand:
                    try:
                        dm.set_plan(plan, par)
                    except RuntimeError:
                        continue
                    res[name] = t()
                best = min(res, key=res.get)
                line = "%8d %6d %9d | %-8s %8.2f | best %-8s %8.2f | %s" % (
                    rows, mean, inf["nnz"], auto_plan, auto_us, best, res[best],
                    " ".join("%s=%.1f" % kv for kv in sorted(res.items())))
                flag = auto_us > 1.10 * res[best]
                print(line + ("   <-- %.0f %% behind (cols=%d)" % ((auto_us / res[best] - 1) * 100, cols) if flag else ""),
                      flush=True)
                if flag:
                    worst.append(line)
                dm.close()
                be.free(dx)
                be.free(dy)
    print("%d shape(s) where AUTO is more than 10 %% behind the best plan" % len(worst))


if __name__ == "__main__":
    main()
