# round 3: SQ counters of the BLOCKED kernels on SparseBench 170^3 (separate --pmc passes, kernel trace only)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3sq
mkdir -p $OUT
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" ; do
  p=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$p -o pmc -- \
      python3 tools/kbench.py --reps 10 --cases sb170 --plans blocked --tunings xcd1nt0 > $OUT/$p.log 2>&1
  echo "pass $p rc=$?"
done
python3 - <<'PY' > $OUT/blocked_sq_counters.json
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("gpurun_out/r3sq/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for k in ("k_blk_products", "k_blk_bins"):
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        d = out.setdefault(k, {})
        for c, v in cs.items():
            d[c] = round(sum(v) / len(v))
            d["dispatches"] = len(v)
for k, d in out.items():
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    d["derived"] = {
        "valu_active_share_of_wave_cycles": round(d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / wc, 4),
        "lds_active_share_of_wave_cycles": round(d.get("SQ_ACTIVE_INST_LDS", 0) * 4 / wc, 4),
        "wait_any_share_of_wave_cycles": round(d.get("SQ_WAIT_ANY", 0) * 4 / wc, 4),
        "lds_bank_conflict_cycles_per_lds_active_cycle": round(d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_ACTIVE_INST_LDS", 1), 1), 4),
    }
json.dump(out, open("/dev/stdout", "w"), indent=1)
PY
cat $OUT/blocked_sq_counters.json | head -70
rm -rf $OUT/*/*.db 2>/dev/null
