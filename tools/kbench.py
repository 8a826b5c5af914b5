#!/usr/bin/env python3
"""Kernel microbenchmark: times every plan / tuning combination on the benchmark shapes with HIP
events (spmv_hip_time_device) and prints achieved algorithmic GB/s.  GPU box only.

    python tools/kbench.py [--reps 50] [--cases C,sb120,rmat20,...]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import matgen  # noqa: E402
import pkgload  # noqa: E402


def build_case(pkg, name, dtype):
    if name in ("S", "W", "A", "B", "C", "D"):
        m = pkg.npb.NpbMatrix(name)
        a = m.a if dtype == np.float64 else m.a.astype(np.float32)
        return m.rowstr, m.colidx, a, m  # keep m alive
    if name.startswith("D/"):  # one block of the class D row partition, e.g. D/8 = part 3 of 8 (x has 1.5 M entries)
        parts = int(name[2:])
        m = pkg.npb.NpbMatrix("D", part=min(3, parts - 1), nparts=parts)
        return m.rowstr, m.colidx, m.a.astype(dtype), m
    if name == "cband":  # class C row lengths and values, but columns consecutive from the diagonal: local gathers
        m = pkg.npb.NpbMatrix("C")
        lens = np.diff(m.rowstr)
        rowid = np.repeat(np.arange(m.na, dtype=np.int64), lens)
        k = np.arange(m.nnz, dtype=np.int64) - np.repeat(m.rowstr[:-1].astype(np.int64) - 1, lens)
        ci = ((rowid + k) % m.na + 1).astype(np.int32)
        return m.rowstr.copy(), ci, m.a.astype(dtype), None
    if name.startswith("sb"):  # the C generator of the SparseBench counterpart (host/callers.c sb_generate): seconds, not half a minute
        m = pkg.callers.sb_generate(int(name[2:]))
        rs, ci, a = m.arrays()
        return rs, ci, a.astype(dtype), m
    if name.startswith("npb:"):  # tools/dispatch_audit.py's grid: npb:<rows>x<mean>, NPB-like row lengths, scattered columns
        rows, mean = (int(v) for v in name[4:].split("x"))
        return matgen.npb_like(7, rows, mean, dtype) + (None,)
    if name.startswith("rmat"):
        return matgen.rmat_graph(20, int(name[4:]), 16, dtype, values="normal") + (None,)
    if name.startswith("grid"):
        n = int(name[4:])
        return matgen.grid2d_graph(n, n, dtype) + (None,)
    raise ValueError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--cases", default="C,sb100,rmat20,grid2000")
    ap.add_argument("--dtypes", default="f64")
    ap.add_argument("--tunings", default="xcd1nt0,xcd0nt0,xcd1nt1")
    ap.add_argument("--wpb", type=int, default=8, help="waves per workgroup of the TILED kernel (8|16)")
    ap.add_argument("--xbytes", type=int, default=32768, help="bytes of x per LDS stripe of the TILED kernel")
    ap.add_argument("--blocks", type=int, default=0, help="TILED workgroups aimed for (0 = default 512)")
    ap.add_argument("--tk", type=int, default=0, help="TILED: 256-entry sub-windows per register window")
    ap.add_argument("--swgs", type=int, default=0, help="STRIPED: workgroups aimed for (0 = the shape's default: 256 / 512)")
    ap.add_argument("--sshape", default="0", help="STRIPED: comma list of kernel shapes to time: 0 = chosen per matrix, 1 = 16 waves x 128 KiB, 2 = 8 waves x 64 KiB (two workgroups per CU)")
    ap.add_argument("--sgroups", default="0", help="STRIPED: comma list of stripe-group counts to time (0 = the builder's choice); "
                                                   "each value builds the layout anew")
    ap.add_argument("--swrows", type=int, default=1024, help="STRIPED: most rows per wave range (multiple of 64, <= 1024)")
    ap.add_argument("--abl", default="0", help="STRIPED: comma list of ablation bit masks to time after the real kernel (1 no fold, "
                                              "2 plain partial stores, 4 no accumulator updates, 8 no ring writes); results are wrong")
    ap.add_argument("--stamps", action="store_true", help="STRIPED: one extra launch that leaves per-workgroup clock stamps; prints "
                                                          "when workgroups start, have x staged and finish")
    ap.add_argument("--ab-rounds", type=int, default=0, help="STRIPED: time the real kernel and the --abl masks in turn, this many rounds")
    ap.add_argument("--stamps-abl", type=int, default=0, help="further tuning-key-6 bits for the stamped launch")
    ap.add_argument("--stamps-launches", type=int, default=0, help="STRIPED: this many further stamped launches; says how much of a "
                    "workgroup's lateness comes back launch after launch (its place on the chip) and how much is chance")
    ap.add_argument("--plans", default="", help="comma list of plan names to run (default all)")
    args = ap.parse_args()
    pkg = pkgload.load_package()
    # ablation masks and clock stamps exist in the debug build only (include/spmv_hip_debug.h); plain timings use the
    # shipped library
    diag = args.stamps or args.ab_rounds > 0 or any(int(v) for v in args.abl.split(","))
    be = pkg.Backend(pkg.lib_path("libhip-spmv-debug.so")) if diag else pkg.Backend()
    if diag:
        print("# debug library: %s" % be.path)
    assert be.device_count() >= 1
    be.set_tuning(2, args.wpb)
    be.set_tuning(3, args.xbytes)
    be.set_tuning(4, args.blocks)
    be.set_tuning(5, args.tk)
    be.set_tuning(9, args.swgs)
    be.set_tuning(12, args.swrows)
    groups = [int(v) for v in args.sgroups.split(",")]
    be.set_tuning(11, groups[0])
    plans = [("striped", pkg.PLAN_STRIPED, 0), ("blocked", pkg.PLAN_BLOCKED, 0), ("hub", pkg.PLAN_HUB, 0), ("tiled", pkg.PLAN_TILED, 0), ("stream", pkg.PLAN_STREAM, 0), ("vec64", pkg.PLAN_VECTOR, 64), ("vec32", pkg.PLAN_VECTOR, 32),
             ("vec16", pkg.PLAN_VECTOR, 16), ("vec8", pkg.PLAN_VECTOR, 8), ("vec4", pkg.PLAN_VECTOR, 4),
             ("vec1", pkg.PLAN_VECTOR, 1), ("binned", pkg.PLAN_BINNED, 0), ("ordered", pkg.PLAN_ORDERED, 0),
             ("auto", pkg.PLAN_AUTO, 0)]
    print("%-10s %-4s %-8s %-9s %10s %10s %8s" % ("case", "dt", "plan", "tuning", "us/spmv", "GB/s(alg)", "GFLOP/s"))
    for dts in args.dtypes.split(","):
        dtype = np.float64 if dts == "f64" else np.float32
        for case in args.cases.split(","):
            t0 = time.time()
            rs, ci, a, keep = build_case(pkg, case, dtype)
            rows = rs.size - 1
            dm = be.matrix(a, rs, ci, rows, cols_hint=(keep.na if case.startswith("D/") else 0))
            inf = dm.info()
            x = np.random.default_rng(1).standard_normal(inf["cols"] + 2).astype(dtype)
            dx = be.malloc(x.nbytes)
            dy = be.malloc(rows * dtype().itemsize + 16)
            be.h2d(dx, x)
            print("# %s %s: rows=%d cols=%d nnz=%d len %d..%d mean %.1f blocks=%d long=%d alg_bytes=%d (setup %.1f s)" % (
                case, dts, rows, inf["cols"], inf["nnz"], inf["min_row_len"], inf["max_row_len"], inf["mean_row_len"],
                inf["n_blocks"], inf["n_long_rows"], inf["algorithmic_bytes"], time.time() - t0), flush=True)
            for tun in args.tunings.split(","):
                xcd, nt = int(tun[3]), int(tun[6])
                be.set_tuning(0, xcd)
                be.set_tuning(1, nt)
                for pname, plan, param in plans:
                    if args.plans and pname not in args.plans.split(","):
                        continue
                    if tun != args.tunings.split(",")[0] and pname in ("ordered", "auto", "vec1", "vec4", "vec8"):
                        continue
                    try:
                        dm.set_plan(plan, param)
                    except RuntimeError as e:
                        print("%-10s %-4s %-8s %-9s   unavailable: %s" % (case, dts, pname, tun, e), flush=True)
                        continue
                    reps = args.reps if pname not in ("ordered", "vec1") else max(3, args.reps // 10)
                    dm.time_device(dx, dy, 3 if pname in ("ordered", "vec1") else 12)  # the first passes over a fresh layout are slower
                    ms = dm.time_device(dx, dy, reps) / reps
                    i2 = dm.info()
                    print("%-10s %-4s %-8s %-9s %10.2f %10.1f %8.1f   launches=%d" % (
                        case, dts, pname, tun, ms * 1e3, i2["algorithmic_bytes"] / ms / 1e6,
                        2.0 * i2["nnz"] / ms / 1e6, i2["n_launches"]), flush=True)
                    if pname == "striped" and args.stamps:
                        import ctypes
                        be.set_tuning(6, 64 | args.stamps_abl)
                        dm.time_device(dx, dy, 1)
                        be.set_tuning(6, 0)
                        nwg = min(i2["n_blocks"], 1024)
                        buf = (ctypes.c_ulonglong * (4 * nwg))()
                        be.lib.spmv_hip_debug_stamps.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
                        assert be.lib.spmv_hip_debug_stamps(0, buf, 4 * nwg) == 0
                        st = np.array(buf[:], dtype=np.float64).reshape(nwg, 4) * 0.01  # us
                        t0 = st[:, 0].min()
                        q = lambda v: "min %.2f med %.2f max %.2f" % (v.min(), np.median(v), v.max())  # noqa: E731
                        print("#   stamps (us after the first workgroup's entry): entry %s | x staged %s | wave 0 done %s | last wave done %s"
                              % (q(st[:, 0] - t0), q(st[:, 1] - t0), q(st[:, 2] - t0), q(st[:, 3] - t0)), flush=True)
                        print("#   staging takes %s us; workgroups end within %.2f us of each other" % (
                            q(st[:, 1] - st[:, 0]), st[:, 3].max() - st[:, 3].min()), flush=True)
                        if args.stamps_launches > 0:
                            ends = []
                            for _ in range(args.stamps_launches):
                                be.set_tuning(6, 64 | args.stamps_abl)
                                dm.time_device(dx, dy, 1)
                                be.set_tuning(6, 0)
                                assert be.lib.spmv_hip_debug_stamps(0, buf, 4 * nwg) == 0
                                s2 = np.array(buf[:], dtype=np.float64).reshape(nwg, 4) * 0.01
                                ends.append(s2[:, 3] - s2[:, 0].min())
                            ends = np.array(ends)                      # [launch, workgroup]
                            rel = ends - ends.mean(axis=1, keepdims=True)  # lateness against the launch's mean end
                            per_wg = rel.mean(axis=0)
                            print("#   %d stamped launches: a workgroup's end against its launch's mean: std over everything %.2f us, "
                                  "std of the per-workgroup means %.2f us (what its place explains), mean launch length %.2f, "
                                  "mean of the last end %.2f, of the median end %.2f" % (
                                      args.stamps_launches, rel.std(), per_wg.std(), ends.max(axis=1).mean(), ends.max(axis=1).mean(),
                                      np.median(ends, axis=1).mean()), flush=True)
                            per_xcd = max(1, (nwg + 7) // 8)
                            xcd = np.arange(nwg) // per_xcd
                            print("#   mean lateness by index // %d (us): %s" % (per_xcd, " ".join("%+.2f" % per_wg[xcd == k].mean() for k in range(8))), flush=True)
                            print("#   mean lateness by index %% 8 (us):  %s" % " ".join("%+.2f" % per_wg[np.arange(nwg) % 8 == k].mean() for k in range(8)), flush=True)
                            print("#   (tuning xcd1: index // %d is the XCD; xcd0: index %% 8 is)" % per_xcd, flush=True)
                            order = np.argsort(per_wg)
                            print("#   earliest workgroups (index: us): %s" % " ".join("%d:%+.1f" % (i, per_wg[i]) for i in order[:8]), flush=True)
                            print("#   latest workgroups   (index: us): %s" % " ".join("%d:%+.1f" % (i, per_wg[i]) for i in order[-8:]), flush=True)
                            if args.stamps_launches >= 40:
                                print("#   per-workgroup mean lateness, in index order: %s" % " ".join("%+.1f" % v for v in per_wg), flush=True)
                    if pname == "striped" and args.ab_rounds > 0:
                        # interleaved A/B (boxes drift by a few per cent within a run): the real kernel and every mask, in turn
                        masks = [0] + [int(v) for v in args.abl.split(",") if int(v)]
                        acc = {mk: [] for mk in masks}
                        for _ in range(args.ab_rounds):
                            for mk in masks:
                                be.set_tuning(6, mk)
                                dm.time_device(dx, dy, 2)
                                acc[mk].append(dm.time_device(dx, dy, 20) / 20 * 1e3)
                        be.set_tuning(6, 0)
                        for mk in masks:
                            v = np.array(acc[mk])
                            print("#   A/B %d rounds x 20 launches, tuning-6 mask %3d: mean %.2f us  median %.2f  std %.2f" % (
                                args.ab_rounds, mk, v.mean(), np.median(v), v.std()), flush=True)
                    elif pname == "striped":
                        for ab in [int(v) for v in args.abl.split(",") if int(v)]:
                            be.set_tuning(6, ab)
                            dm.time_device(dx, dy, 3)
                            ms = dm.time_device(dx, dy, reps) / reps
                            be.set_tuning(6, 0)
                            print("%-10s %-4s %-8s %-9s %10.2f   (ablation %d: timing only)" % (case, dts, pname, tun, ms * 1e3, ab), flush=True)
            be.set_tuning(0, 1)
            be.set_tuning(1, 0)
            dm.close()
            for ng in groups[1:]:  # the STRIPED layout again with other stripe groups (a layout is built once per matrix object)
                be.set_tuning(11, ng)
                dm = be.matrix(a, rs, ci, rows, cols_hint=(keep.na if case.startswith("D/") else 0))
                try:
                    dm.set_plan(pkg.PLAN_STRIPED, 0)
                    dm.time_device(dx, dy, 3)
                    ms = dm.time_device(dx, dy, args.reps) / args.reps
                    i2 = dm.info()
                    print("%-10s %-4s %-8s %-9s %10.2f %10.1f %8.1f   workgroups=%d" % (
                        case, dts, "striped", "groups=%d" % ng, ms * 1e3, i2["algorithmic_bytes"] / ms / 1e6,
                        2.0 * i2["nnz"] / ms / 1e6, i2["n_blocks"]), flush=True)
                except RuntimeError as e:
                    print("%-10s %-4s %-8s groups=%d unavailable: %s" % (case, dts, "striped", ng, e), flush=True)
                dm.close()
            be.set_tuning(11, groups[0])
            be.free(dx)
            be.free(dy)
            if case.startswith("sb") and keep is not None:
                del rs, ci, a
                pkg.callers.free(keep)
            del keep


if __name__ == "__main__":
    main()
