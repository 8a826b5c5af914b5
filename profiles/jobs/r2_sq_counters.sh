# round 2: SQ counters of k_striped, class C and a class D row block in separate processes (separate --pmc passes, kernel trace only)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2c/sq
mkdir -p $OUT
for case in C D/8; do
  tag=$(echo $case | tr -d /)
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
              "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" ; do
    p=$(echo $pass | cut -d' ' -f1)
    timeout -k 10 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/${tag}_$p -o pmc -- \
        python3 tools/kbench.py --reps 10 --cases $case --plans striped --tunings xcd1nt0 > $OUT/${tag}_$p.log 2>&1
    echo "$case pass $p rc=$?"
  done
done
python3 - <<'PY' > gpurun_out/r2c/striped_sq_counters.json
import csv, glob, collections, json, os
out = {}
for f in sorted(glob.glob("gpurun_out/r2c/sq/*/*counter_collection.csv")):
    case = f.split("/")[-2].split("_")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_striped" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = out.setdefault(case, {})
    for c, v in acc.items():
        d[c] = round(sum(v) / len(v))
        d["dispatches"] = len(v)
for case, d in out.items():
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    d["derived"] = {
        "valu_active_share_of_wave_cycles": round(d.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4),
        "any_inst_active_share": round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4),
        "issue_stall_share (SQ_WAIT_INST_ANY)": round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 4),
        "waitcnt_or_barrier_share (SQ_WAIT_ANY)": round(d.get("SQ_WAIT_ANY", 0) / wc, 4),
        "valu_insts_per_wave": round(d.get("SQ_INSTS_VALU", 0) / max(1, d.get("SQ_WAVES", 1)), 1),
        "lds_bank_conflict_share_of_lds_cycles": round(d.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, d.get("SQ_LDS_IDX_ACTIVE", 1)), 4),
        "note": "per-wave shares; four waves share a SIMD, so a VALU share of 0.15 is a SIMD's vector pipe 60 % busy",
    }
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r2c/striped_sq_counters.json | head -60
rm -rf $OUT/*/*.db
