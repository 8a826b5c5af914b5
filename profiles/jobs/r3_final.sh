# round 3: the GPU suite + smoke, the default bench line, its rocprofv3 kernel trace and PMC traffic, clean vs instrumented k_striped
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3k
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
timeout -k 10 600 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --no-cpu --no-configs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "trace rc=$?"
python3 tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/bench_kernel_by_grid.csv; head -12 $O/bench_kernel_by_grid.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python3 bench.py --no-cpu --no-npb --no-configs --steps 20 --warmup 2 > $O/pmc_$c.json 2> $O/pmc_$c.err; echo "pmc $c rc=$?"
done
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE/pmc_counter_collection.csv $O/pmc_WRITE_SIZE/pmc_counter_collection.csv > $O/pmc_traffic.json; head -c 1500 $O/pmc_traffic.json
( echo "# k_striped on NPB class C: the shipped instantiation (DIAG = false; tuning-6 mask 0) against the instrumented twin with every feature on"
  echo "# (DIAG = true; mask 256 = a bit no code tests); debug library; interleaved A/B"
  timeout -k 10 300 python3 tools/kbench.py --reps 40 --cases C --plans striped --tunings xcd1nt0 --abl 256 --ab-rounds 12 ) > $O/clean_vs_instrumented.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/sq -o sq -- python3 tools/kbench.py --reps 5 --cases C --plans striped --tunings xcd1nt0 --abl 256 > $O/sq.log 2>&1; echo "sq rc=$?"
python3 - <<'PY' >> $O/clean_vs_instrumented.log
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/r3k/sq/sq_counter_collection.csv")):
    if "k_striped" in r["Kernel_Name"]:
        acc["instrumented (DIAG = true)" if "true>" in r["Kernel_Name"].replace(" ", "") or ", true" in r["Kernel_Name"] else "shipped (DIAG = false)"][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per launch (rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES):")
for k, v in acc.items():
    print("#   %-28s %s" % (k, "  ".join("%s %.3e" % (c, sum(x) / len(x)) for c, x in sorted(v.items()))))
PY
cat $O/clean_vs_instrumented.log | tail -8
timeout -k 10 500 python3 tools/dispatch_audit.py > $O/dispatch_audit.log 2>&1; echo "audit rc=$?"; grep "behind" $O/dispatch_audit.log
rm -rf $O/trace/*.db $O/pmc_*/*.db $O/sq/*.db 2>/dev/null
