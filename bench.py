#!/usr/bin/env python3
"""bench.py -- headline measurement of the hip-spmv backend (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path: one y = A*x (FP64 CSR) over the whole workload with A, x
and y already resident in HBM.

  N = 1   workload = BASELINE.json configs[1]: the NPB3.3.1 CG class C matrix (150000 rows,
          36121058 nonzeros), generated in-process by our restatement of makea.
  N > 1   workload = BASELINE.json configs[4]: the NPB CG class D matrix (1500000 rows,
          ~694.4 M nonzeros) row-partitioned over N GPUs (block rows of about equal nnz, x replicated
          on every GPU, NO collective on the data path).  Two ways to get there:
            * under a launcher (WORLD_SIZE = N: one process per GPU, what the driver starts): each rank
              generates only its own row block; torch.distributed is used for the barrier and the
              max-over-ranks of the elapsed time, nothing else;
            * WITHOUT a launcher (`python bench.py --gpus N`): ONE process drives the N GPUs through the
              library's own multi-device paths -- one device-resident row block per GPU for the kernel
              figure (spmv_hip_time_multi), SPMV_HIP_DEVICES=N behind spmv_harness_ for everything that
              goes through the ABI.  With fewer than N GPUs visible the row blocks wrap around the
              visible ones (a rehearsal: the line says so in `devices_used`).
          Either way the line carries, besides `value` (kernels alone, x / y resident in HBM: the
          contract's figure), `gather_inclusive` (what libspmv/gpu.c:264,285 do around the kernels on
          every call: x host -> every GPU, y slices back: the host gather) and `npb_cg`: the complete
          NPB CG class D benchmark through spmv_harness_ from ONE process with SPMV_HIP_DEVICES=N
          (the call pattern of NPB3.3.1/CG/cg.f:532,628), MOp/s by cg.f:395-402, verified against
          cg.f:122-163.

roofline.traffic is measured by the run (N = 1): two child processes of this script behind `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` (`--no-traffic` quotes the stored figure of profiles/traffic.json instead).

value = 2 * nnz * K / elapsed  (effective GFLOP/s, whole job).  The same K launches are
bracketed by HIP events recorded on the stream the kernels run on (spmv_hip_time_device), which
gives the per-launch duration for the roofline object.  Rank 0 at N = 1 additionally reports
  * cold: the first 20 launches after the upload, before anything has settled;
  * npb_cg / npb_cg_device: the complete NPB CG class C benchmark through the drop-in ABI (host vectors,
    PCIe round trip of x and y on each of its 1976 SpMV calls) and through the opt-in device-resident CG;
  * sparsebench_170, graphs_rmat20, parboil: BASELINE configs[2] and [3] on the same library, each
    checked against numpy / the oracle outside its timed region;
  * scale_ref: class D (the N > 1 workload) on this one GPU, the like-for-like first point of the
    strong-scaling series;
  * cpu_baseline: the reference's CPU path (oracle/_ref/native.so when it was built in the
    authoring container, else our port) on one host core, on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# what a kernel that only READS reaches with k_striped's launch and load shapes (256 workgroups x 16 waves, per-wave streams,
# non-temporal 16-byte loads, four register sets in flight) on class C's 366-MB packed stream: tools/ntstream.hip,
# profiles/r04_ntstream_read_ceilings.log (7397 / 7401 GB/s in two runs; 868 MB: 6906 / 6955; 6.9 GB: 6581 / 6687)
MEASURED_READ_CEILING_GBS = 7400.0
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
ZETA_D = 52.514532105794  # cg.f:122-163


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def tolerance(np, lens, absum, eps=2.0 ** -53):
    """the per-row bar of tests/test_gpu_parity.py (assert_close_rows), derived per plan in tests/rowtol.py"""
    from rowtol import row_tolerance
    return row_tolerance(lens, absum, eps)


def check_rows(np, m_a, m_rowstr, m_colidx, x, yv, what):
    """EVERY row of a result against numpy (segment sums of a * x[col]), within the any-order summation bound"""
    prod = m_a * x[m_colidx - 1]
    starts = (m_rowstr[:-1] - m_rowstr[0]).astype(np.int64)
    lens = np.diff(m_rowstr)
    nz = lens > 0
    want = np.zeros(lens.size)
    scale = np.zeros(lens.size)
    if prod.size:
        want[nz] = np.add.reduceat(prod, starts[nz])
        scale[nz] = np.add.reduceat(np.abs(prod), starts[nz])
    badrows = np.nonzero(~(np.abs(yv - want) <= tolerance(np, lens, scale) + 1e-300))[0]
    assert badrows.size == 0, "%s: %d rows differ, first %d: %r vs %r" % (what, badrows.size, badrows[0], yv[badrows[0]], want[badrows[0]])


PRODUCT_KERNELS = ("k_striped", "k_group_sum", "k_stream", "k_long", "k_long_finish", "k_vector", "k_tiled", "k_ordered",
                   "k_blk_products", "k_blk_bins", "k_blk_join", "k_hub", "k_hub_pack", "k_hub_fix")


def traffic_child(args):
    """`bench.py --traffic-child`: the headline matrix, device-resident, 3 + 10 products and nothing else -- the program the
    parent puts behind `rocprofv3 --pmc <counter> --kernel-trace --` (one counter per pass, as the guide prescribes)."""
    import numpy as np
    import pkgload
    pkg = pkgload.load_package()
    be = pkg.Backend()
    m = pkg.npb.NpbMatrix(args.cls or "C")
    dm = be.matrix(m.a, m.rowstr, m.colidx, m.rows, device=0, cols_hint=m.na)
    x = np.random.default_rng(12345).standard_normal(m.na + 2)
    dx, dy = be.malloc(x.nbytes, 0), be.malloc(max(m.rows, 1) * 8, 0)
    be.h2d(dx, x, 0)
    dm.time_device(dx, dy, 3)
    dm.time_device(dx, dy, TRAFFIC_LAUNCHES)
    dm.close()
    be.free(dx, 0), be.free(dy, 0)
    be.release()  # before the interpreter exits: the profiler shuts down first otherwise


TRAFFIC_LAUNCHES = 10


def measure_traffic(cls, launches_per_step):
    """HBM-side bytes per SpMV of the headline workload, measured by THIS run: two child processes of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, kernel trace only; /opt/skills/guides/MI355X_MICROARCH.md,
    HBM section: FETCH_SIZE counts KB and sees half of a wide coalesced read on gfx950 -> x 2; WRITE_SIZE is exact), the last
    TRAFFIC_LAUNCHES products of each.  Returns (bytes with the guide's x 2, detail) or (None, why)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "no rocprofv3 on PATH"
    if any("rocprof" in v.lower() for k, v in os.environ.items() if k in ("LD_PRELOAD", "HSA_TOOLS_LIB")) or \
            any("ROCPROF" in k or k.startswith("ROCP_") for k in os.environ):
        return None, "this process runs under a profiler itself"
    pat = re.compile(r"\b(" + "|".join(PRODUCT_KERNELS) + r")\b")
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--traffic-child", "--cls", cls]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp"))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "%s pass failed (rc %d): %s" % (counter, r.returncode, r.stdout[-300:])
            per = {}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter and pat.search(row["Kernel_Name"]):
                    per[int(row["Dispatch_Id"])] = per.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"]) * 1024.0
            ids = sorted(per)[-TRAFFIC_LAUNCHES * launches_per_step:]
            if len(ids) < TRAFFIC_LAUNCHES * launches_per_step:
                return None, "%s pass: %d product dispatches found, %d expected" % (counter, len(ids), TRAFFIC_LAUNCHES * launches_per_step)
            got[counter] = sum(per[i] for i in ids) / TRAFFIC_LAUNCHES
        except Exception as e:  # a measurement aid must not take the bench line down
            return None, "%s pass: %r" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]), {
        "kind": "measured by this run", "how": "two child processes of bench.py (--traffic-child) under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                "--kernel-trace, the last %d products of each; traffic = 2 x FETCH_SIZE + WRITE_SIZE per product (the guide's gfx950 correction for "
                "wide reads)" % TRAFFIC_LAUNCHES,
        "fetch_size_bytes_as_counted": int(got["FETCH_SIZE"]), "write_size_bytes": int(got["WRITE_SIZE"]),
        "with_round_1_calibration_x2.12": int(2.12 * got["FETCH_SIZE"] + got["WRITE_SIZE"])}


def npb_cg_object(m, be, res, st, runs, cls, path):
    return {
        "class": cls, "mops": round(res["mops"], 2), "time_s": round(res["t_bench"], 4),
        "repeats": len(runs), "mops_all": [round(r["mops"], 1) for r in runs],
        "verified_all": all(bool(r["verified"]) for r in runs),
        "time_in_spmv_s": round(res["t_spmv"], 4), "spmv_calls": res["spmv_calls"],
        "zeta": res["zeta"], "err": res["err"], "verified": bool(res["verified"]),
        "path": path,
        "upload_s": round(st["t_upload"], 4), "init_s": round(st["t_init"], 4),
        # SURVEY 8(d) "also report": the PCIe-inclusive time of one ABI call (entry -> return: x H2D, kernels, y D2H)
        "us_per_abi_call": round(res["t_spmv"] / max(1, res["spmv_calls"]) * 1e6, 2),
        "one_time": "init_s (HIP context + code object, first call of the process) and upload_s (matrix H2D + re-layout, "
                    "inside the untimed first conj_grad, cg.f:233) are not part of time_s",
    }


def npb_through_the_abi(pkg, be, cls, n_dev, repeats, np):
    """The complete NPB CG benchmark through spmv_harness_ from THIS process, the matrix row-partitioned over n_dev GPUs by
    the library itself (SPMV_HIP_DEVICES, read at the first device product of the ABI): what cg.f:532,628 would get."""
    visible = be.device_count()
    if n_dev > 1:
        os.environ["SPMV_HIP_DEVICES"] = str(min(n_dev, visible))
        os.environ["SPMV_HIP_PARTS"] = str(n_dev)
    t0 = time.time()
    m = pkg.npb.NpbMatrix(cls)
    t_gen = time.time() - t0
    be.reset_stats()
    runs = [m.run_cg(be.harness_fn()) for _ in range(max(1, repeats))]
    st = be.stats()
    res = sorted(runs, key=lambda r: r["mops"])[len(runs) // 2]  # median run (reference convention: 5 repeats)
    inf = be.abi_matrix_info()
    obj = npb_cg_object(m, be, res, st, runs, cls,
                        "spmv_harness_ ABI: host x/y every call (PCIe round trip), host vector loops on 1 core" +
                        ("; one process, SPMV_HIP_DEVICES=%d SPMV_HIP_PARTS=%d: x to every GPU, y slices gathered on the host, no collective"
                         % (min(n_dev, visible), n_dev) if n_dev > 1 else ""))
    obj["generated_s"] = round(t_gen, 1)
    obj["uploads"] = int(st["uploads"])
    obj["plan"] = pkg.abi.PLAN_NAMES[inf["plan"]] if inf else None
    if n_dev > 1:
        obj["devices_used"], obj["row_partitions"] = min(n_dev, visible), n_dev
    be.release()
    return obj, m


def other_configs(pkg, be, np, line):
    """BASELINE configs[2] and [3] on the library the headline ran on: SparseBench 170^3 (unsymmetric, BiCG call pattern),
    R-MAT scale 20 (bfs / pagerank call patterns), parboil's two bundled datasets (float ABI, 50 products).  Each checked
    outside its timed region."""
    C = pkg.callers
    # ---- SparseBench 170^3: kernel (HBM-resident) and the ABI call (78 MB of x + y across PCIe)
    t0 = time.time()
    sb = C.sb_generate(170)
    rs, ci, a = sb.arrays()
    dm = be.matrix(a, rs, ci, sb.rows)
    inf = dm.info()
    x = np.random.default_rng(170).standard_normal(inf["cols"] + 2)
    dx, dy = be.malloc(x.nbytes), be.malloc(sb.rows * 8 + 16)
    be.h2d(dx, x)
    dm.time_device(dx, dy, 5)
    ms = dm.time_device(dx, dy, 20) / 20
    y = be.d2h(np.empty(sb.rows), dy)
    check_rows(np, a, rs, ci, x, y, "sparsebench 170 kernel")
    dm.set_plan(pkg.PLAN_STREAM, 0)  # the CSR row-segment kernel on the same matrix, for the record
    dm.time_device(dx, dy, 3)
    ms_stream = dm.time_device(dx, dy, 10) / 10
    dm.close(), be.free(dx), be.free(dy)
    be.reset_stats()
    yy = np.empty(sb.rows)
    be.harness(yy, a, x, rs, ci, sb.rows)  # upload
    t1 = time.perf_counter()
    for _ in range(10):
        be.harness(yy, a, x, rs, ci, sb.rows)
    abi_ms = (time.perf_counter() - t1) / 10 * 1e3
    check_rows(np, a, rs, ci, x, yy, "sparsebench 170 through the ABI")
    be.release()
    line["sparsebench_170"] = {
        "rows": sb.rows, "nnz": int(sb.nnz), "plan": pkg.abi.PLAN_NAMES[inf["plan"]], "launches_per_product": inf["n_launches"],
        "kernel_us": round(ms * 1e3, 2), "frac": round(inf["algorithmic_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "kernel_us_stream_plan": round(ms_stream * 1e3, 2),
        "algorithmic_bytes_per_launch": inf["algorithmic_bytes"], "abi_ms_per_call": round(abi_ms, 3),
        "abi_moves": "x and y, 39 MB each, host <-> GPU on every call (random.f:32,72)", "setup_s": round(time.time() - t0, 1),
    }
    C.free(sb)
    # ---- graphs: R-MAT 2^20, edge factor 16: bfs (16 sources, values 1) and pagerank (column-normalised x 0.85)
    t0 = time.time()
    g = C.graph_rmat(20, 16)
    rs, ci, a = g.arrays()
    be.reset_stats()
    t1 = time.perf_counter()
    levels = 0
    cand = np.nonzero(np.diff(rs) >= 8)[0]
    sources = cand[:: max(1, cand.size // 16)][:16]  # 16 vertices that have somewhere to go
    for src in sources:
        lv, dist = C.bfs_run(g, be.harness_fn(), int(src))
        levels += lv
    bfs_s = time.perf_counter() - t1
    st = be.stats()
    bfs_attr = {"bfs_time_in_spmv_s": round(st["t_total"], 4), "bfs_upload_s": round(st["t_upload"], 4),
                "bfs_abi_ms_per_call": round((st["t_total"] - st["t_upload"]) / max(1, st["calls"]) * 1e3, 4)}
    be.release()
    dm = be.matrix(a, rs, ci, g.rows)
    inf = dm.info()
    x = np.random.default_rng(20).standard_normal(inf["cols"] + 2)
    dx, dy = be.malloc(x.nbytes), be.malloc(g.rows * 8 + 16)
    be.h2d(dx, x)
    dm.time_device(dx, dy, 5)
    ms = dm.time_device(dx, dy, 20) / 20
    check_rows(np, a, rs, ci, x, be.d2h(np.empty(g.rows), dy), "rmat 20 kernel")
    dm.close(), be.free(dx), be.free(dy)
    iters = 20
    be.reset_stats()
    t1 = time.perf_counter()
    pr, err = C.pagerank(g, be.harness_fn(), iters)
    pr_ms = (time.perf_counter() - t1) / iters * 1e3
    st = be.stats()
    pr_attr = {"pagerank_time_in_spmv_s": round(st["t_total"], 4), "pagerank_upload_s": round(st["t_upload"], 4),
               "pagerank_abi_ms_per_call": round((st["t_total"] - st["t_upload"]) / max(1, st["calls"]) * 1e3, 4),
               "pagerank_spmv_calls": int(st["calls"])}
    assert np.all(np.isfinite(pr)) and np.all(pr >= 0.0) and 0.0 < float(pr.sum()) <= 1.0 + 1e-9  # dangling vertices leak rank (main.cpp:128-149 does not redistribute it)
    be.release()
    line["graphs_rmat20"] = dict(bfs_attr, **pr_attr)
    line["graphs_rmat20"].update({
        "rows": g.rows, "nnz": int(g.nnz), "plan": pkg.abi.PLAN_NAMES[inf["plan"]], "spmv_us": round(ms * 1e3, 2),
        "frac": round(inf["algorithmic_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "bfs_s": round(bfs_s, 4), "bfs_sources": int(len(sources)), "bfs_spmv_calls": int(levels), "bfs_uploads": int(st["uploads"]),
        "pagerank_ms_per_iter": round(pr_ms, 3), "pagerank_iters": iters, "setup_s": round(time.time() - t0, 1),
        "attribution": "x_time_in_spmv_s = seconds between entry to and return from spmv_harness_ (the library's own clock, the one-time upload "
                       "included); x_abi_ms_per_call = the same without the upload, per call: what is left "
                       "of bfs_s and pagerank_ms_per_iter is the caller counterpart's own loops (host/callers.c)",
        "what": "bfs/bfs.cc:36-68 and pagerank/main.cpp:128-149 call patterns through spmv_harness_ (first call uploads); "
                "frontiers are bit-exact by construction (integer-valued products), checked in tests/test_gpu_parity.py",
    })
    C.free(g)
    # ---- parboil: the two bundled datasets, 50 products each through f_spmv_harness_ under the default dispatch
    pb = {}
    for name, mtx in (("small", "1138_bus.mtx"), ("medium", "bcsstk18.mtx")):
        d = os.path.join(ROOT, "tests", "golden", "parboil_" + name)
        if not os.path.exists(os.path.join(d, mtx)):
            continue
        m = C.pb_load(os.path.join(d, mtx))
        be.reset_stats()
        t1 = time.perf_counter()
        yv = C.pb_run(m, be.harness_fn(np.float32), os.path.join(d, "vector.bin"), 50)
        pb[name + "_ms"] = round((time.perf_counter() - t1) * 1e3, 3)
        st = be.stats()
        pb[name + "_dispatched_to"] = "host loop" if st["host_calls"] == st["calls"] else ("device" if st["host_calls"] == 0 else "both")
        assert np.all(np.isfinite(yv))
        be.release()
        C.free(m)
    if pb:
        pb["what"] = "parboil/benchmarks/spmv/src/cpu/main.c:93-95: 50 x f_spmv_harness_, file reading excluded; outputs pass the " \
                     "reference's comparator in tests/test_gpu_parity.py"
        pb["dispatch"] = os.environ.get("SPMV_HIP_DISPATCH", "auto")
        line["parboil"] = pb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cls", default=None, help="override the NPB class (default: C at N=1, D at N>1)")
    ap.add_argument("--plan", default="auto", help="auto|stream|vector:L|binned|ordered|tiled|striped|blocked|hub")
    ap.add_argument("--prewarm", type=int, default=600, help="untimed launches before the W warm-up steps (a FIXED count: the first "
                    "hundreds of launches over a fresh layout run 1-3 us slower, profiles/r03_coldstart_first_300_launches.log); "
                    "the figure after the W warm-up steps alone is reported beside the headline in `after_warmup_only`")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-npb", action="store_true", help="skip the full NPB CG runs through the ABI")
    ap.add_argument("--no-configs", action="store_true", help="skip the sparsebench / graphs / parboil objects (N = 1)")
    ap.add_argument("--cpu-iters", type=int, default=10, help="timed CG iterations of the CPU sample")
    ap.add_argument("--no-scale-ref", action="store_true", help="skip the class D single-GPU reference point at N=1")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic (two rocprofv3 --pmc child runs, ~30 s); "
                    "the stored figure of profiles/traffic.json is quoted instead")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--scale-ref-cls", default=None, help="class of the scale_ref object (default: D, the N > 1 workload, when the "
                    "headline is class C; naming a class runs it whatever the headline's class -- the tests use B)")
    ap.add_argument("--npb-repeats", type=int, default=5, help="runs of the full NPB CG benchmark at N=1; the median is reported")
    args = ap.parse_args()
    if args.traffic_child:
        return traffic_child(args)

    import numpy as np
    import torch
    import pkgload
    sys.path.insert(0, os.path.join(ROOT, "tests"))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # no launcher: ONE process drives --gpus N GPUs through the library's own multi-device paths (never a silent N = 1)
    inproc = world == 1 and args.gpus > 1
    if world > 1 and world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    n_gpus = args.gpus if inproc else world
    dist = None
    wait_group = None
    # rehearsal aid: BENCH_DIST_BACKEND=gloo lets several ranks share the one GPU of a test box
    # (RCCL refuses two ranks on one device); the driver's real runs use nccl = RCCL, one GPU per rank
    backend_name = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank % max(ndev, 1))
        if backend_name == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend_name)
        # the ranks that only WAIT while rank 0 runs the one-process NPB legs on all GPUs do so on the CPU: an RCCL barrier is a
        # kernel spinning on every GPU, and those GPUs are the ones rank 0's row blocks run on
        wait_group = dist.new_group(backend="gloo") if backend_name == "nccl" else None

    pkg = pkgload.load_package()
    if not os.path.exists(pkg.lib_path()):
        pkg.build_all()
    be = pkg.Backend()
    if be.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libhip-spmv.so has no CPU fallback")
    device = (local_rank % max(ndev, 1)) if world > 1 else 0
    torch.cuda.set_device(device)
    red_dev = "cuda" if backend_name == "nccl" else "cpu"
    cls = args.cls or ("C" if n_gpus == 1 else "D")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # =============================================================== the kernel figure
    t0 = time.time()
    cold = None
    after_warmup_only = None
    gather = None
    rank_ms = None
    devices_used = None
    if inproc:
        # ---- one process, N device-resident row blocks (wrapping onto the visible GPUs when there are fewer)
        visible = be.device_count()
        devices_used = min(n_gpus, visible)
        parts = [pkg.npb.NpbMatrix(cls, part=p, nparts=n_gpus) for p in range(n_gpus)]
        log("class %s in %d row blocks generated in %.1f s" % (cls, n_gpus, time.time() - t0))
        m = parts[0]
        x = np.random.default_rng(12345).standard_normal(m.na + 2)
        dms, dxs, dys = [], [], []
        for p, mp in enumerate(parts):
            d = p % visible
            dms.append(be.matrix(mp.a, mp.rowstr, mp.colidx, mp.rows, device=d, cols_hint=mp.na))
            dxs.append(be.malloc(x.nbytes, d))
            dys.append(be.malloc(max(mp.rows, 1) * 8, d))
            be.h2d(dxs[-1], x, d)
        info = dms[0].info()
        infos = [dm.info() for dm in dms]
        PREWARM = 0
        for _ in range(max(12, args.prewarm // 6) // 12):  # a fixed count here too (class D row blocks: fewer, longer launches)
            be.time_multi(dms, dxs, dys, 12)
            PREWARM += 12
        be.time_multi(dms, dxs, dys, max(1, args.warmup))
        t_start = time.perf_counter()
        ev_each = be.time_multi(dms, dxs, dys, args.steps)  # K launches per block, rep by rep; HIP events on every block's stream
        elapsed = time.perf_counter() - t_start
        ev_ms = max(ev_each)
        rank_ms = [round(v / args.steps, 6) for v in ev_each]
        nnz_total = float(sum(mp.nnz for mp in parts))
        for p, mp in enumerate(parts):
            y = be.d2h(np.empty(mp.rows), dys[p], p % visible)
            check_rows(np, mp.a, mp.rowstr, mp.colidx, x, y, "kernel, row block %d" % p)
        alg_bytes_rank0 = info["algorithmic_bytes"]
        for p in range(n_gpus):
            dms[p].close()
            be.free(dxs[p], p % visible)
            be.free(dys[p], p % visible)
        for mp in parts:
            mp.close()
        m_na, m_rows_total = m.na, sum(i["rows"] for i in infos)
    else:
        if n_gpus == 1:
            m = pkg.npb.NpbMatrix(cls)
        else:
            m = pkg.npb.NpbMatrix(cls, part=rank, nparts=n_gpus)
        log("[rank %d] class %s rows [%d,%d) nnz=%d generated in %.1f s" % (
            rank, cls, m.row_begin, m.row_end, m.nnz, time.time() - t0))
        # device-resident block, replicated x, local y
        dm = be.matrix(m.a, m.rowstr, m.colidx, m.rows, device=device, cols_hint=m.na)
        if args.plan != "auto":
            name, _, par = args.plan.partition(":")
            plan = {"stream": pkg.PLAN_STREAM, "vector": pkg.PLAN_VECTOR, "binned": pkg.PLAN_BINNED,
                    "ordered": pkg.PLAN_ORDERED, "tiled": pkg.PLAN_TILED, "striped": pkg.PLAN_STRIPED,
                    "blocked": pkg.PLAN_BLOCKED, "hub": pkg.PLAN_HUB}[name]
            dm.set_plan(plan, int(par) if par else 0)
        info = dm.info()
        x = np.random.default_rng(12345).standard_normal(m.na + 2)  # same on every rank: x is replicated
        dx = be.malloc(x.nbytes, device)
        dy = be.malloc(max(m.rows, 1) * 8, device)
        be.h2d(dx, x, device)
        # cold: the first 20 launches after the upload, before clocks, TLBs and caches have seen the layout
        cold_ms = dm.time_device(dx, dy, 20) / 20
        cold = {"launch_us": round(cold_ms * 1e3, 3), "launches": 20,
                "achieved": round(info["algorithmic_bytes"] / (cold_ms * 1e-3) / 1e9, 2),
                "frac": round(info["algorithmic_bytes"] / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "what": "the first 20 launches after the upload of the matrix, nothing run before them (DESIGN.md 4.4: "
                        "the first launches over a fresh layout run slower)"}
        # what the command line alone asks for: W warm-up steps, then K timed ones (the 20 cold launches came before them)
        for _ in range(args.warmup):
            dm.spmv_device(dx, dy)
        be.synchronize(device)
        aw_ms = dm.time_device(dx, dy, args.steps) / args.steps
        after_warmup_only = {"launch_us": round(aw_ms * 1e3, 3), "value": round(2.0 * m.nnz / (aw_ms * 1e-3) / 1e9, 3), "unit": "GFLOP/s",
                             "frac": round(info["algorithmic_bytes"] / (aw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "what": "this rank's K steps timed right after the 20 cold launches and the W warm-up steps, before the fixed pre-warm"}
        # untimed: a FIXED pre-warm (--prewarm launches), then the W warm-up steps again; the line reports both figures
        PREWARM = 20 + args.warmup + args.steps
        for _ in range(max(0, args.prewarm) // 12):
            dm.time_device(dx, dy, 12)
            PREWARM += 12
        for _ in range(args.warmup):
            dm.spmv_device(dx, dy)
        be.synchronize(device)
        barrier()
        t_start = time.perf_counter()
        ev_ms = dm.time_device(dx, dy, args.steps)  # K launches, HIP events on the launch stream, synchronised
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t_start
        barrier()
        nnz_total = float(m.nnz)
        if dist is not None:
            mine = torch.zeros(n_gpus, dtype=torch.float64, device=red_dev)
            mine[rank] = elapsed * 1e3 / args.steps
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)  # every rank's own ms per step, for the balance of the partition
            rank_ms = [round(float(v), 6) for v in mine.tolist()]
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            s = torch.tensor([float(m.nnz)], dtype=torch.float64, device=red_dev)
            dist.all_reduce(s, op=dist.ReduceOp.SUM)
            nnz_total = float(s.item())
        # the timed kernel really computed y = A*x: EVERY local row against numpy, outside the timed region
        y = np.empty(m.rows)
        be.d2h(y, dy, device)
        check_rows(np, m.a, m.rowstr, m.colidx, x, y, "kernel")
        alg_bytes_rank0 = info["algorithmic_bytes"]
        m_na = m.na

        # N > 1 under a launcher: the same step with the traffic of an ABI call around the kernels (libspmv/gpu.c:264,285)
        if dist is not None:
            from lilac_benchmarks_amd.hostgather import SharedVector
            tag = "hipspmv_bench_%s" % os.environ.get("MASTER_PORT", "0")
            if rank == 0:
                xs = SharedVector(tag + "_x", m.na + 2, create=True)
                xs.arr[:] = x
                xs.arr.flush()
                SharedVector(tag + "_y", m.na, create=True).close()
            barrier()
            if rank != 0:
                xs = SharedVector(tag + "_x", m.na + 2)
            ys = SharedVector(tag + "_y", m.na)
            pinned_x = be.host_register(xs.arr)
            pinned_y = be.host_register(ys.arr)
            y_slice = ys.arr[m.row_begin:m.row_end]
            dm.time_call(xs.arr, y_slice, dx, dy, max(1, args.warmup))
            barrier()
            t_start = time.perf_counter()
            ev_ms_g = dm.time_call(xs.arr, y_slice, dx, dy, args.steps)
            torch.cuda.synchronize()
            el_g = time.perf_counter() - t_start
            barrier()
            mine = torch.zeros(n_gpus, dtype=torch.float64, device=red_dev)
            mine[rank] = el_g * 1e3 / args.steps
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
            tg = torch.tensor([el_g], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            check_rows(np, m.a, m.rowstr, m.colidx, x, np.array(y_slice), "host-gathered slice")  # what landed in the shared host vector
            barrier()
            if rank == 0:
                assert bool(np.all(np.isfinite(ys.arr)))  # the caller's view: the whole y, gathered with no collective
            gather = {
                "value": round(2.0 * nnz_total * args.steps / float(tg.item()) / 1e9, 3), "unit": "GFLOP/s",
                "ms_per_step": round(float(tg.item()) * 1e3 / args.steps, 6),
                "rank_ms_per_step": [round(float(v), 6) for v in mine.tolist()],
                "rank0_events_ms_per_step": round(ev_ms_g / args.steps, 6),
                "x_bytes_per_rank_per_step": int(m.na * 8), "y_bytes_this_rank_per_step": int(m.rows * 8),
                "pinned": bool(pinned_x and pinned_y),
                "what": "per step and rank: x H2D from one pinned shared host vector (libspmv/gpu.c:264), the kernels, the rank's y "
                        "slice D2H straight to its row offset of one shared host vector (gpu.c:285; the host gather); no collective",
            }
            if pinned_x:
                be.host_unregister(xs.arr)
            if pinned_y:
                be.host_unregister(ys.arr)
            del y_slice
            xs.close(unlink=(rank == 0))
            ys.close(unlink=(rank == 0))

    ms_per_step = elapsed * 1e3 / args.steps
    gflops = 2.0 * nnz_total * args.steps / elapsed / 1e9
    launch_ms = ev_ms / args.steps
    achieved = alg_bytes_rank0 / (launch_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if n_gpus == 1 and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = "%s/%s" % (cls, pkg.abi.PLAN_NAMES[info["plan"]])
            ent = tj.get(key, {})
            traffic = ent.get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_src = {"kind": "stored PMC figure, not measured by this run", "file": "profiles/traffic.json",
                               "collected_from": ent.get("source"), "fetch_correction": ent.get("fetch_correction"),
                               "guide_correction": 2.0}
        except Exception:
            traffic = None
    if rank == 0 and n_gpus == 1 and not args.no_traffic and args.plan == "auto":
        live, detail = measure_traffic(cls, info["n_launches"])
        if live is not None:
            if traffic_src is not None:
                detail["stored_figure_for_comparison"] = {"hbm_bytes_per_launch": traffic, "file": traffic_src.get("file"),
                                                          "fetch_correction": traffic_src.get("fetch_correction")}
            traffic, traffic_src = live, detail
        elif traffic_src is not None:
            traffic_src["not_measured_because"] = detail
    na_total = m_na
    line = {
        "metric": "NPB CG class %s SpMV effective GFLOP/s (FP64 CSR y=A*x, HBM-resident; %% of HBM roofline "
                  "in `roofline`, NPB CG MOp/s through the ABI in `npb_cg`)" % cls,
        "value": round(gflops, 3),
        "unit": "GFLOP/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 6),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": ("NPB3.3.1 CG class %s matrix (literal restatement of cg.f's makea; pinned by nnz, row-length extrema, "
                         "y[0] and the NPB zeta verification, tests/test_npb.py), "
                         "na=%d, nnz=%d, one y=A*x per step, A/x/y resident in HBM" % (cls, na_total, int(nnz_total))),
            "class": cls,
            "rows": na_total,
            "nnz": int(nnz_total),
            "plan": pkg.abi.PLAN_NAMES[info["plan"]],
            "plan_param": info["plan_param"],
            "launches_per_step": info["n_launches"],
            "parallelism": "1 GPU" if n_gpus == 1 else
                           "block-row partition over %d GPUs (equal nnz), x replicated, no collective" % n_gpus,
            "processes": ("one process driving %d GPUs through the library (no launcher)" % n_gpus) if inproc else
                         ("one process per GPU (torch.distributed, %s)" % backend_name if n_gpus > 1 else "one process"),
            "rank_ms_per_step": rank_ms,
            "untimed_steps_before_timing": PREWARM + args.warmup,
            "prewarm": "%d untimed steps (the 20 `cold` launches, the steps of `after_warmup_only`, then a fixed %d launches) before the "
                       "%d warm-up steps; the figures without them are in `cold` and `after_warmup_only`" % (PREWARM, args.prewarm, args.warmup),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": "k_%s" % pkg.abi.PLAN_NAMES[info["plan"]],
            "launches_per_step": info["n_launches"],
            "algorithmic_bytes_per_launch": alg_bytes_rank0,
            "launch_us": round(launch_ms * 1e3, 3),
            "scope": "the slowest row block (its own algorithmic bytes over its own launch time)" if n_gpus > 1 else "whole matrix",
            # context, not the contract's fraction: what a kernel that only READS memory reaches on an MI355X
            "measured_read_ceiling": {"GB/s": MEASURED_READ_CEILING_GBS, "frac_of_it": round(achieved / MEASURED_READ_CEILING_GBS, 4),
                                      "of": "algorithmic bytes; the kernel's own stream is 10 B per nonzero where the algorithmic count is 12",
                                      "source": "profiles/r04_ntstream_read_ceilings.log (stored figure: a read-only kernel with "
                                                "k_striped's launch and load shapes on a stream of class C's size)"},
        },
    }
    if cold is not None:
        line["cold"] = cold
    if after_warmup_only is not None:
        line["after_warmup_only"] = after_warmup_only
    if devices_used is not None:
        line["config"]["devices_used"] = devices_used
        if devices_used < n_gpus:
            line["config"]["rehearsal"] = "%d row blocks wrapped onto %d visible GPU(s): the code path of an N-GPU run, not its speed" % (n_gpus, devices_used)
    if gather is not None:
        line["gather_inclusive"] = gather
    if n_gpus > 1:
        line["config"]["value_is"] = "kernels only, x / y resident in HBM (the contract's figure); gather_inclusive adds the per-call x H2D and y D2H"

    # =============================================================== N > 1: the metric's first term, NPB CG MOp/s
    if n_gpus > 1:
        if not inproc:
            dm.close()
            be.free(dx, device)
            be.free(dy, device)
            m.close()
            barrier()  # every rank has freed its row block
        if rank == 0 and not args.no_npb:
            obj, mfull = npb_through_the_abi(pkg, be, cls, n_gpus, 1, np)
            line["npb_cg"] = obj
            # the same benchmark with conj_grad resident on the GPUs: one row block per GPU, the scalars through pinned host
            # memory, the slices of p peer to peer (spmv_hip_cg_create_multi); no collective library
            vis = be.device_count()
            rdm = mfull.run_cg_device_parts(be, n_gpus, min(n_gpus, vis))
            line["npb_cg_device"] = {
                "class": cls, "mops": round(rdm["mops"], 2), "time_s": round(rdm["t_bench"], 4), "zeta": rdm["zeta"], "err": rdm["err"],
                "verified": bool(rdm["verified"]), "row_partitions": n_gpus, "devices_used": min(n_gpus, vis),
                "path": "spmv_hip_cg_create_multi: every GPU keeps its row block of A, its slices of x, z, q, r and a full copy of p; per CG "
                        "iteration two scalars cross through pinned host memory (added in part order) and every part pushes its slice of p "
                        "to the others with hipMemcpyPeerAsync; one host thread issues everything",
            }
            if inproc:
                # the gather-inclusive step, one process: K ABI calls on the full matrix (x to every GPU, kernels, y slices back)
                xx = np.random.default_rng(12345).standard_normal(mfull.na + 2)
                yy = np.empty(mfull.na)
                be.harness(yy, mfull.a, xx, mfull.rowstr, mfull.colidx, mfull.na)  # upload
                k = max(3, min(args.steps, 20))
                t1 = time.perf_counter()
                for _ in range(k):
                    be.harness(yy, mfull.a, xx, mfull.rowstr, mfull.colidx, mfull.na)
                el = (time.perf_counter() - t1) / k
                check_rows(np, mfull.a, mfull.rowstr, mfull.colidx, xx, yy, "ABI call over %d row partitions" % n_gpus)
                line["gather_inclusive"] = {
                    "value": round(2.0 * mfull.nnz / el / 1e9, 3), "unit": "GFLOP/s", "ms_per_step": round(el * 1e3, 6), "steps": k,
                    "what": "one spmv_harness_ call per step from this process (SPMV_HIP_DEVICES): x from pageable host memory into a "
                            "pinned buffer and from there to every GPU at once (libspmv/gpu.c:264), the kernels, every GPU's y "
                            "slice back to its row offset of the caller's y (gpu.c:285: the host gather); no collective",
                }
                be.release()
            mfull.close()
        if dist is not None:
            dist.barrier(group=wait_group) if wait_group is not None else dist.barrier()

    if rank == 0 and n_gpus == 1 and not args.no_npb:
        be.reset_stats()
        runs = [m.run_cg(be.harness_fn()) for _ in range(max(1, args.npb_repeats))]
        st = be.stats()
        res = sorted(runs, key=lambda r: r["mops"])[len(runs) // 2]  # median run (reference convention: 5 repeats)
        line["npb_cg"] = npb_cg_object(m, be, res, st, runs, cls,
                                       "spmv_harness_ ABI: host x/y every call (PCIe round trip), host vector loops on 1 core")
        be.release()
        # the same benchmark with the opt-in device-resident CG step (vectors never leave the GPU)
        runsd = [m.run_cg_device(be) for _ in range(max(1, args.npb_repeats))]
        resd = sorted(runsd, key=lambda r: r["mops"])[len(runsd) // 2]
        line["npb_cg_device"] = {
            "class": cls, "mops": round(resd["mops"], 2), "time_s": round(resd["t_bench"], 4),
            "repeats": len(runsd), "mops_all": [round(r["mops"], 1) for r in runsd],
            "verified_all": all(bool(r["verified"]) for r in runsd),
            "zeta": resd["zeta"], "err": resd["err"], "verified": bool(resd["verified"]),
            "path": "spmv_hip_cg_* opt-in entry points: p,q,r,z,x resident in HBM, 4 kernels per CG iteration (p.q folded into the SpMV), "
                    "host sees two norms per outer iteration",
        }

    if rank == 0 and n_gpus == 1 and not args.no_npb:
        # the FP32 twin of the ABI (f_spmv_harness_, what parboil calls) on the same matrix: reported beside the
        # FP64 headline, never instead of it
        a32 = m.a.astype(np.float32)
        dm32 = be.matrix(a32, m.rowstr, m.colidx, m.rows, device=device, cols_hint=m.na)
        inf32 = dm32.info()
        x32 = x.astype(np.float32)
        dx32 = be.malloc(x32.nbytes, device)
        dy32 = be.malloc(max(m.rows, 1) * 4 + 16, device)
        be.h2d(dx32, x32, device)
        dm32.time_device(dx32, dy32, 10)
        ms32 = dm32.time_device(dx32, dy32, 50) / 50
        line["spmv_f32"] = {
            "value": round(2.0 * m.nnz / (ms32 * 1e-3) / 1e9, 3), "unit": "GFLOP/s", "us_per_spmv": round(ms32 * 1e3, 3),
            "plan": pkg.abi.PLAN_NAMES[inf32["plan"]], "algorithmic_bytes_per_launch": inf32["algorithmic_bytes"],
            "roofline_frac": round(inf32["algorithmic_bytes"] / (ms32 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "FP32 values + 16-bit columns are 6 B per nonzero (217 MB for class C): less than the 256-MB Infinity "
                    "Cache, so a repeated product is partly cache-resident",
        }
        dm32.close()
        be.free(dx32, device)
        be.free(dy32, device)

    if rank == 0 and n_gpus == 1 and not args.no_configs:
        other_configs(pkg, be, np, line)

    sr_cls = args.scale_ref_cls or ("D" if cls == "C" else None)
    if rank == 0 and n_gpus == 1 and sr_cls and not args.no_scale_ref:
        # the workload of the N > 1 runs (class D, BASELINE configs[4]) on this one GPU, whole matrix:
        # the like-for-like 1-GPU point for the strong-scaling series (the headline above stays class C)
        t0 = time.time()
        md = pkg.npb.NpbMatrix(sr_cls)
        t_gen = time.time() - t0
        t0 = time.time()
        dmd = be.matrix(md.a, md.rowstr, md.colidx, md.rows, device=device, cols_hint=md.na)
        t_up = time.time() - t0
        infd = dmd.info()
        xd = np.random.default_rng(12345).standard_normal(md.na + 2)
        dxd = be.malloc(xd.nbytes, device)
        dyd = be.malloc(md.rows * 8, device)
        be.h2d(dxd, xd, device)
        cold_d = dmd.time_device(dxd, dyd, 20) / 20
        dmd.time_device(dxd, dyd, 20)  # untimed: the first passes over a freshly uploaded 7-GB layout run 5-10 % slower
        ksteps = 20
        msd = dmd.time_device(dxd, dyd, ksteps) / ksteps
        line["scale_ref"] = {
            "workload": "NPB CG class %s, whole matrix on 1 GPU (what the N > 1 runs partition)" % sr_cls, "class": sr_cls,
            "value": round(2.0 * md.nnz / (msd * 1e-3) / 1e9, 3), "unit": "GFLOP/s", "ms_per_step": round(msd, 6),
            "steps": ksteps, "warmup": 40, "nnz": int(md.nnz), "plan": pkg.abi.PLAN_NAMES[infd["plan"]],
            "roofline_frac": round(infd["algorithmic_bytes"] / (msd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "cold_ms_per_step": round(cold_d, 6),
            "algorithmic_bytes_per_launch": infd["algorithmic_bytes"], "generated_s": round(t_gen, 1), "upload_s": round(t_up, 2),
            "launches_per_step": infd["n_launches"], "workgroups": infd["n_blocks"],
        }
        try:  # stored PMC figure of this workload (profiles/traffic.json), not measured by this run
            ent = json.load(open(tpath)).get("%s/%s" % (sr_cls, pkg.abi.PLAN_NAMES[infd["plan"]]), {})
            line["scale_ref"]["traffic"] = ent.get("hbm_bytes_per_launch")
            line["scale_ref"]["traffic_source"] = ent.get("source")
        except Exception:
            pass
        dmd.close()
        be.free(dxd, device)
        be.free(dyd, device)
        if not args.no_npb:
            # the first point of the MOp/s series (the metric at N > 1 is NPB CG class D): the complete benchmark on this one
            # GPU, through the drop-in ABI and through the device-resident CG (cg.f:395-402; zeta against cg.f:122-163)
            be.reset_stats()
            rd = md.run_cg(be.harness_fn())
            std = be.stats()
            line["scale_ref"]["npb_cg"] = npb_cg_object(md, be, rd, std, [rd], sr_cls,
                                                        "spmv_harness_ ABI: host x/y every call (PCIe round trip), host vector loops on 1 core")
            if sr_cls == "D":
                line["scale_ref"]["npb_cg"]["zeta_reference"] = ZETA_D
            be.release()
            rdd = md.run_cg_device(be)
            line["scale_ref"]["npb_cg_device"] = {
                "class": sr_cls, "mops": round(rdd["mops"], 2), "time_s": round(rdd["t_bench"], 4), "zeta": rdd["zeta"], "err": rdd["err"],
                "verified": bool(rdd["verified"]),
                "path": "spmv_hip_cg_* opt-in entry points: vectors resident in HBM, 4 kernels per CG iteration",
            }
        md.close()

    if rank == 0 and n_gpus == 1 and not args.no_cpu:
        from oracle import oracle as orc  # checker / baseline only
        if orc.have_reference():
            cpu, kind = orc.Reference(), "reference"
        else:
            cpu, kind = orc.Oracle(), "port"
        res = m.run_cg(cpu.harness_fn(), niter=args.cpu_iters)
        calls = res["spmv_calls"]
        timed_calls = 26 * args.cpu_iters
        line["cpu_baseline"] = {
            "value": round(2.0 * m.nnz * timed_calls / res["t_spmv"] / 1e9, 4),
            "unit": "GFLOP/s",
            "cores": 1,
            "kind": kind,
            "sample": "%d of the 1976 SpMV calls of NPB CG class %s (%d untimed + %d timed CG iterations) on the "
                      "same matrix through the same spmv_harness_ ABI; %s, gcc -std=c99 -O3, 1 thread of %d (%s)" % (
                          calls, cls, 1, args.cpu_iters,
                          "reference libspmv/native.c" if kind == "reference" else "oracle/spmv_oracle.c",
                          os.cpu_count() or 0, cpu_model()),
            "npb_mops": round(res["mops"], 2),
            "ms_per_spmv": round(res["t_spmv"] / timed_calls * 1e3, 3),
        }
        # context only (optional): the same loop row-parallel on this box's CPU share -- same bits,
        # labelled separately from the faithful single-core baseline above
        share = min(16, os.cpu_count() or 1)
        port = orc.Oracle()
        ymt = np.empty(m.rows)
        port.spmv_mt(m.a, x[: m.na].copy(), m.rowstr, m.colidx, m.rows, share, out=ymt)  # warm
        reps_mt = 40
        t0 = time.perf_counter()
        for _ in range(reps_mt):
            _, used = port.spmv_mt(m.a, x[: m.na], m.rowstr, m.colidx, m.rows, share, out=ymt)
        t_mt = (time.perf_counter() - t0) / reps_mt
        line["cpu_baseline_threads"] = {
            "value": round(2.0 * m.nnz / t_mt / 1e9, 3), "unit": "GFLOP/s", "cores": int(used), "kind": "port",
            "ms_per_spmv": round(t_mt * 1e3, 3),
            "sample": "%d products of the same matrix with oracle/spmv_oracle_mt.c: native-impl.c's loop on %d threads, "
                      "rows split by nonzeros (bit-identical to the single-core result)" % (reps_mt, int(used)),
        }

    if n_gpus == 1:
        dm.close()
        be.free(dx, device)
        be.free(dy, device)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
