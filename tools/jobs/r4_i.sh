# round 4: the whole GPU suite on the current tree, then bench.py plain and under rocprofv3 (kernel trace + stats)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4i
mkdir -p $OUT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/tests_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"; tail -5 $OUT/tests_all.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "rocprof bench rc=$?"
python3 tools/trace_summary.py $OUT/trace/*kernel_trace.csv > $OUT/kernel_by_grid.csv 2>/dev/null; head -40 $OUT/kernel_by_grid.csv
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +30M -delete
echo done
