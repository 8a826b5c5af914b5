#!/usr/bin/env python3
"""The device-resident CG over N row partitions (spmv_hip_cg_create_multi), one host thread against a thread per part: NPB CG of a
class, N parts wrapped onto the visible GPUs.  On one GPU the kernels of the parts share the card, so the figure of interest is
the HOST side: a small class is bound by the API calls per iteration, which is what 8 GPUs would see.
    python3 tools/cgparts.py [--cls A] [--parts 1,2,4,8]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pkgload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cls", default="A")
ap.add_argument("--parts", default="1,2,4,8")
args = ap.parse_args()
pkg = pkgload.load_package()
be = pkg.Backend()
m = pkg.npb.NpbMatrix(args.cls)
one = m.run_cg_device(be)
print("class %s: one GPU, one partition: %.4f s = %.0f MOp/s  zeta %.13f" % (args.cls, one["t_bench"], one["mops"], one["zeta"]), flush=True)
for n in [int(v) for v in args.parts.split(",")]:
    for mode in ("0", "1"):
        os.environ["SPMV_HIP_CG_THREADS"] = mode
        r = m.run_cg_device_parts(be, n)
        it = 26 * m.niter
        print("  %d partitions on %d GPU(s), %s: %.4f s = %.0f MOp/s, %.1f us per CG iteration, verified %d, |zeta - one| %.1e" % (
            n, r["devices_used"], "a host thread per part" if mode == "1" else "one host thread     ", r["t_bench"], r["mops"],
            r["t_bench"] / it * 1e6, r["verified"], abs(r["zeta"] - one["zeta"])), flush=True)
