cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3l
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
/usr/bin/time -v timeout -k 10 600 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"; grep "Elapsed (wall" $O/bench_line.err
timeout -k 10 500 python3 tools/dispatch_audit.py > $O/dispatch_audit.log 2>&1; echo "audit rc=$?"; grep "behind" $O/dispatch_audit.log
( echo "# the reference's own programs (compiled from its sources, oracle/_ref) on native.so and on libhip-spmv.so under SPMV_HIP_DISPATCH=auto"
  python3 tools/ref_callers_compare.py ) > $O/ref_programs.log 2>&1; tail -15 $O/ref_programs.log
