# round 4, final set from one box with the final code: the whole GPU suite, smoke(), bench.py plain and under rocprofv3, two --pmc passes for the traffic of class C / D
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4final
mkdir -p $OUT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/tests_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"; tail -4 $OUT/tests_all.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-700 $OUT/bench.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "rocprof bench rc=$?"
python3 tools/trace_summary.py $OUT/trace/*kernel_trace.csv > $OUT/kernel_by_grid.csv 2>/dev/null; head -12 $OUT/kernel_by_grid.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python3 bench.py --no-cpu --no-npb --no-configs --steps 20 --warmup 2 --prewarm 24 > $OUT/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/pmc_traffic.json 2>$OUT/pmc_traffic.err; head -c 1500 $OUT/pmc_traffic.json
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
echo done
