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
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)  # KB -> bytes
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def short(name):
    for k in ("k_striped", "k_tiled", "k_stream", "k_long"):
        if k in name:
            return k
    return None


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"kernels": {}, "hbm_bytes_per_spmv": 0.0,
           "correction": "FETCH_SIZE x2.12 for the 16-byte + 8-byte load mix of k_striped / k_tiled, x2.00 "
                         "otherwise (gfx950 counts half of wide reads; calibrated in round 1); WRITE_SIZE exact"}
    total = 0.0
    for name, (fb, n) in sorted(fetch.items()):
        k = short(name)
        if not k:
            continue
        wb = write.get(name, (0.0, 0))[0]
        corr = FETCH_CORRECTION.get(k, FETCH_CORRECTION["default"])
        out["kernels"][name[:90]] = {"dispatches": n, "FETCH_SIZE_bytes": fb, "WRITE_SIZE_bytes": wb,
                                     "fetch_correction": corr, "hbm_bytes": fb * corr + wb}
        total += fb * corr + wb
    out["hbm_bytes_per_spmv"] = total
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
