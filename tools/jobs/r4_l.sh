# round 4: (1) k_striped with fewer accumulators and deeper rings on class C (clean debug instantiations, wave ranges <= 256 rows);
#          (2) the dispatch audit again after HUB's criterion went back to the LDS tier alone
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4l
mkdir -p $OUT
python3 - > $OUT/rings.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np, pkgload
from kbench import build_case
pkg = pkgload.load_package()
be = pkg.Backend(pkg.lib_path("libhip-spmv-debug.so"))
for case in ("C", "B"):
    rs, ci, a, keep = build_case(pkg, case, np.float64)
    rows = rs.size - 1
    be.set_tuning(12, 256)  # wave ranges of at most 256 rows: 4 accumulators suffice
    dm = be.matrix(a, rs, ci, rows)
    dm.set_plan(pkg.PLAN_STRIPED, 0)
    inf = dm.info()
    x = np.random.default_rng(1).standard_normal(inf["cols"] + 2)
    dx, dy = be.malloc(x.nbytes), be.malloc(rows * 8 + 16)
    be.h2d(dx, x)
    be.set_tuning(5, 0)
    y0 = dm.spmv_host(x)
    variants = [(0, "2 sets, 16 accumulators (shipped)"), (12, "2 sets, 4 accumulators"), (13, "3 sets, 8 accumulators"), (14, "4 sets, 4 accumulators")]
    acc = {k: [] for k, _ in variants}
    same = {}
    for k, _ in variants:
        be.set_tuning(5, k)
        same[k] = bool(np.array_equal(dm.spmv_host(x).view(np.uint64), y0.view(np.uint64)))
    for _ in range(8):
        for k, _ in variants:
            be.set_tuning(5, k)
            dm.time_device(dx, dy, 3)
            acc[k].append(dm.time_device(dx, dy, 20) / 20 * 1e3)
    be.set_tuning(5, 0)
    print("# class %s, STRIPED with wave ranges <= 256 rows (%d workgroups), 8 interleaved rounds x 20 launches" % (case, inf["n_blocks"]), flush=True)
    for k, name in variants:
        v = np.array(acc[k])
        print("%-36s mean %7.2f us  median %7.2f  min %7.2f  std %.2f   same bits as the shipped kernel: %s" % (name, v.mean(), np.median(v), v.min(), v.std(), same[k]), flush=True)
    dm.close(); be.free(dx); be.free(dy)
PY
echo "rings rc=$?"; cat $OUT/rings.log
timeout -k 10 900 python3 tools/dispatch_audit.py > $OUT/audit.log 2>&1; echo "audit rc=$?"; grep -c "behind" $OUT/audit.log; tail -2 $OUT/audit.log
timeout -k 10 600 python3 tools/dispatch_audit.py --f32 > $OUT/audit_f32.log 2>&1; echo "audit f32 rc=$?"; tail -2 $OUT/audit_f32.log
