#!/usr/bin/env python3
"""tools/pmc_traffic.py -- HBM bytes per SpMV launch from two rocprofv3 --pmc passes.

    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o pmc -- \
            python3 bench.py --no-cpu --no-npb --no-scale-ref --steps 20 --warmup 2
    done
    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv \
                                gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv  > profiles/...json

Counters are collected in their own passes (never together with --sys-trace etc.), as
/opt/skills/guides/MI355X_MICROARCH.md prescribes.  Units: FETCH_SIZE / WRITE_SIZE count kilobytes
per dispatch; on gfx950 FETCH_SIZE sees half of wide coalesced reads (the guide's HBM section), which
profiles/r01_bench_classC_striped_pmc_traffic.json (round 1) confirmed on known byte counts: x2.00 for
16-byte-per-lane reads, x2.12 for k_striped's 16-byte + 8-byte mix.  WRITE_SIZE is exact.
"""
import csv
import json
import sys
from collections import defaultdict

FETCH_CORRECTION = {"k_striped": 2.12, "k_tiled": 2.12, "default": 2.0}


def per_kernel(path, counter):
    """counter bytes per dispatch, per kernel, in dispatch order"""
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[r["Kernel_Name"]].append((int(r.get("Dispatch_Id", len(acc[r["Kernel_Name"]]))), float(r["Counter_Value"]) * 1024.0))
    return {k: [b for _, b in sorted(v)] for k, v in acc.items()}  # KB -> bytes


def workloads(values):
    """dispatch indices grouped by value: values more than a factor of 1.5 apart are different matrices (class C and
    class D launches of k_striped share one grid); groups in ascending order of value"""
    order = sorted(range(len(values)), key=lambda i: values[i])
    groups, cur = [], [order[0]]
    for i in order[1:]:
        if values[i] > 1.5 * max(values[cur[-1]], 1.0):
            groups.append(cur)
            cur = []
        cur.append(i)
    groups.append(cur)
    return groups


def short(name):
    for k in ("k_striped", "k_group_sum", "k_tiled", "k_stream", "k_long", "k_blk_products", "k_blk_bins"):
        if k in name:
            return k
    return None


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"kernels": {}, "hbm_bytes_per_spmv": {},
           "correction": "FETCH_SIZE x2.12 for the 16-byte + 8-byte load mix of k_striped / k_tiled, x2.00 "
                         "otherwise (gfx950 counts half of wide reads: the guide says x2, round 1 calibrated x2.12 for "
                         "this mix on known byte counts); WRITE_SIZE exact",
           "workloads": "dispatches of one kernel are grouped by counter value (factor 1.5 apart = another matrix): "
                        "#0 the smaller matrix (class C), #1 the larger (class D)"}
    totals = defaultdict(float)
    for name, fvals in sorted(fetch.items()):
        k = short(name)
        if not k:
            continue
        wvals = write.get(name, [])
        corr = FETCH_CORRECTION.get(k, FETCH_CORRECTION["default"])
        for tag, idx in enumerate(workloads(fvals)):  # both passes ran the same command: dispatch i is dispatch i
            fb = sum(fvals[i] for i in idx) / len(idx)
            wsel = [wvals[i] for i in idx if i < len(wvals)]
            wb = sum(wsel) / len(wsel) if wsel else 0.0
            out["kernels"]["%s #%d" % (k, tag)] = {"dispatches": len(idx), "FETCH_SIZE_bytes": fb, "WRITE_SIZE_bytes": wb,
                                                  "fetch_correction": corr, "hbm_bytes": fb * corr + wb,
                                                  "hbm_bytes_guide_x2": fb * 2.0 + wb}
            totals[str(tag)] += fb * corr + wb
    out["hbm_bytes_per_spmv"] = {("workload #%s" % t): v for t, v in sorted(totals.items())}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
