# round 2: SQ counters of k_striped on class C and on a class D row block (separate --pmc passes, kernel trace only)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2a/sq
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" ; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- \
      python3 tools/kbench.py --reps 5 --cases C,D/8 --plans striped --tunings xcd1nt0 > $OUT/$tag.log 2>&1
  echo "pass $tag rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "gpurun_out/r2a/sq")
for f in sorted(glob.glob(out + "/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_striped" not in k and "k_group_sum" not in k:
            continue
        key = (k[:40], r.get("Grid_Size", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in acc.items():
        print(f.split("/")[-2], key, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
