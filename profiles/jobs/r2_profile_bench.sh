# round 2: rocprofv3 of the bench command (kernel trace + stats), then HBM traffic from two --pmc passes of a short bench
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2a/prof
mkdir -p $OUT
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "trace rc=$?"
python3 tools/trace_summary.py $OUT/trace/bench_kernel_trace.csv > $OUT/bench_kernel_by_grid.csv; head -12 $OUT/bench_kernel_by_grid.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python3 bench.py --no-cpu --no-npb --steps 20 --warmup 2 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err; echo "pmc $c rc=$?"
done
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/pmc_traffic.json; cat $OUT/pmc_traffic.json
rm -rf $OUT/trace/*.db $OUT/pmc_*/*.db 2>/dev/null
ls -la $OUT $OUT/trace | head -30
