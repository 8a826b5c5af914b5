# round 3: the whole GPU suite + smoke, the default bench line, the PATHSAMPLE counterpart (value refreshes), SparseBench through the ABI, the dispatch audit
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3g
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
( echo "# spmvdrv pathsample <lib> <nmin> 100: per-call times of PFOLD (the a-position array rewritten before every call) and NGT (nothing changes)"
  for n in 40000 400000; do
    for mode in auto device; do echo "== nmin=$n libhip-spmv.so SPMV_HIP_DISPATCH=$mode"; SPMV_HIP_DISPATCH=$mode SPMV_HIP_STATS=1 lilac-benchmarks_amd/lib/spmvdrv pathsample lilac-benchmarks_amd/lib/libhip-spmv.so $n 100; done
    echo "== nmin=$n the reference's native.so"; lilac-benchmarks_amd/lib/spmvdrv pathsample oracle/_ref/native.so $n 100; done
  echo "== nmin=400000, 1000 calls, device"; SPMV_HIP_DISPATCH=device SPMV_HIP_STATS=1 lilac-benchmarks_amd/lib/spmvdrv pathsample lilac-benchmarks_amd/lib/libhip-spmv.so 400000 1000
) > $O/pathsample_driver.log 2>&1
cat $O/pathsample_driver.log
( for ov in 1 0; do echo "== spmvdrv sb 170, SPMV_HIP_YOVERLAP=$ov"; SPMV_HIP_YOVERLAP=$ov SPMV_HIP_STATS=1 lilac-benchmarks_amd/lib/spmvdrv sb lilac-benchmarks_amd/lib/libhip-spmv.so 170; done ) > $O/sb170_driver.log 2>&1
cat $O/sb170_driver.log
timeout -k 10 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
tail -c 1500 $O/bench_n1.json; tail -3 $O/bench_n1.err
timeout -k 10 500 python3 tools/dispatch_audit.py > $O/dispatch_audit.log 2>&1; echo "audit rc=$?"; tail -6 $O/dispatch_audit.log
