# round 4: the committed bench line (final code: roofline.traffic measured by the run) and the same command under rocprofv3 for the kernel stats
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4final3
mkdir -p $OUT
(time timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -4 $OUT/bench.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "rocprof bench rc=$?"
python3 tools/trace_summary.py $OUT/trace/*kernel_trace.csv > $OUT/kernel_by_grid.csv 2>/dev/null; head -5 $OUT/kernel_by_grid.csv
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
