#!/usr/bin/env python3
"""Folds rocprofv3 --pmc counter_collection CSVs (one directory per pass) into one JSON: per case, kernel and grid size the
mean of every SQ counter over the dispatches, plus the shares the DESIGN quotes.
    python tools/sq_summary.py <dir with pass subdirectories> <kernels, comma separated> [case,case,...]
Kernels are matched by their exact name (k_hub does not match k_hub_fix).  With a case list, the dispatches are cut
into cases at every upload of a matrix (k_colrange runs once per spmv_hip_matrix_create: tools/kbench.py creates one
matrix per case, in the order given)."""
import collections
import csv
import glob
import json
import re
import sys

NAME = re.compile(r"\b(k_[a-z0-9_]+)\b")


def main():
    root, want = sys.argv[1], set(sys.argv[2].split(","))
    cases = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    out = {}
    for f in sorted(glob.glob(root + "/*/*counter_collection.csv") + glob.glob(root + "/*/*/*counter_collection.csv")):
        rows = list(csv.DictReader(open(f)))
        marks = sorted({int(r["Dispatch_Id"]) for r in rows if "k_colrange" in r["Kernel_Name"]})
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            m = NAME.search(r["Kernel_Name"])
            if not m or m.group(1) not in want:
                continue
            case = ""
            if cases:
                n = sum(1 for d in marks if d <= int(r["Dispatch_Id"]))
                case = (cases[n - 1] if 1 <= n <= len(cases) else "case%d" % n) + " "
            acc["%s%s grid=%s" % (case, m.group(1), r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            d = out.setdefault(k, {})
            for c, v in cs.items():
                d[c] = round(sum(v) / len(v))
                d["dispatches"] = len(v)
    for k, d in out.items():
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        d["derived"] = {
            "valu_active_share_of_wave_cycles": round(d.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4),
            "lds_active_share_of_wave_cycles": round(d.get("SQ_ACTIVE_INST_LDS", 0) / wc, 4),
            "wait_any_share_of_wave_cycles (s_waitcnt / barrier)": round(d.get("SQ_WAIT_ANY", 0) / wc, 4),
            "wait_inst_any_share_of_wave_cycles (waiting to issue)": round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 4),
            "wait_inst_lds_share_of_wave_cycles": round(d.get("SQ_WAIT_INST_LDS", 0) / wc, 4),
            "valu_insts_per_wave": round(d.get("SQ_INSTS_VALU", 0) / max(1, d.get("SQ_WAVES", 1)), 1),
            "vmem_rd_insts_per_wave": round(d.get("SQ_INSTS_VMEM_RD", 0) / max(1, d.get("SQ_WAVES", 1)), 1),
            "lds_insts_per_wave": round(d.get("SQ_INSTS_LDS", 0) / max(1, d.get("SQ_WAVES", 1)), 1),
            "lds_bank_conflict_cycles_per_lds_active_cycle": round(d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_ACTIVE_INST_LDS", 1), 1), 4),
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
