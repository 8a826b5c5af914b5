# round 4: where do class C's 69 us go?  
#  - timing-only ablations of k_striped on C and B (debug library): no scan (16), no scan / ring / accumulators (28),
#    one-row-per-lane emulation (32: loads + LDS gathers + 4 adds per lane and sub-window)
#  - register rings of 3 and 4 sets with the non-temporal stream loads (debug twins, tuning key 5)
#  - clock stamps of 40 launches of the final kernel; rocprofv3 kernel trace of bench.py for the gap and k_group_sum
#  - the read ceiling from the Infinity Cache (streams that fit it, ordinary loads)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4g
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/tests_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"; tail -5 $OUT/tests_all.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/kbench.py --cases C,B --plans striped --tunings xcd1nt0 --ab-rounds 6 --abl 16,28,32 > $OUT/abl.log 2>&1; echo "abl rc=$?"; cat $OUT/abl.log
for tk in 3 4; do
  timeout -k 10 200 python3 tools/kbench.py --cases C --plans striped --tunings xcd1nt0 --tk $tk --ab-rounds 6 --abl 256 > $OUT/ring$tk.log 2>&1; echo "ring $tk rc=$?"; grep "A/B" $OUT/ring$tk.log
done
timeout -k 10 200 python3 tools/kbench.py --cases C --plans striped --tunings xcd1nt0 --ab-rounds 6 --abl 256 > $OUT/ring2.log 2>&1; grep "A/B" $OUT/ring2.log
timeout -k 10 300 python3 tools/kbench.py --cases C --plans striped --tunings xcd1nt0 --stamps --stamps-launches 40 > $OUT/stamps.log 2>&1; echo "stamps rc=$?"; cat $OUT/stamps.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 20 --warmup 5 > $OUT/bench_under_rocprof.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench_under_rocprof.log | cut -c1-600
python3 tools/trace_summary.py $OUT/trace/*kernel_trace.csv > $OUT/kernel_by_grid.csv 2>/dev/null; head -20 $OUT/kernel_by_grid.csv
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +30M -delete
echo done
