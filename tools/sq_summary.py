#!/usr/bin/env python3
"""Folds rocprofv3 --pmc counter_collection CSVs (one directory per pass) into one JSON: per kernel name fragment and per
grid size (a case), the mean of every SQ counter over the dispatches, plus the shares the DESIGN quotes.
    python tools/sq_summary.py <dir with pass subdirectories> <kernel fragments, comma separated>"""
import collections
import csv
import glob
import json
import sys


def main():
    root, frags = sys.argv[1], sys.argv[2].split(",")
    out = {}
    for f in sorted(glob.glob(root + "/*/*counter_collection.csv") + glob.glob(root + "/*/*/*counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for k in frags:
                if k in r["Kernel_Name"] and (k != "k_long" or "finish" not in r["Kernel_Name"]):
                    acc["%s grid=%s" % (k, r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
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
            "lds_insts_per_wave": round(d.get("SQ_INSTS_LDS", 0) / max(1, d.get("SQ_WAVES", 1)), 1),
            "lds_bank_conflict_cycles_per_lds_active_cycle": round(d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_ACTIVE_INST_LDS", 1), 1), 4),
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
