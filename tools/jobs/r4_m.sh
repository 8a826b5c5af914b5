# round 4: (1) where an ABI call of the pagerank / bfs counterparts spends its time, by pinning mode (SPMV_HIP_STATS=2);
#          (2) the dispatch audit in f32 and f64 after "hubs means hubs"; (3) hub tests
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4m
mkdir -p $OUT
L=lilac-benchmarks_amd/lib
for pin in auto staging pageable; do
  echo "== pagerank rmat:20:16, 40 iterations, SPMV_HIP_PIN=$pin" >> $OUT/pin.log
  SPMV_HIP_PIN=$pin SPMV_HIP_STATS=2 timeout -k 10 120 $L/spmvdrv pagerank $L/libhip-spmv.so rmat:20:16 40 1 >> $OUT/pin.log 2>&1
  echo "== bfs rmat:20:16, 16 sources, SPMV_HIP_PIN=$pin" >> $OUT/pin.log
  SPMV_HIP_PIN=$pin SPMV_HIP_STATS=2 timeout -k 10 120 $L/spmvdrv bfs $L/libhip-spmv.so rmat:20:16 16 >> $OUT/pin.log 2>&1
done
cat $OUT/pin.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "hub or auto_takes or graph" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout -k 10 600 python3 tools/dispatch_audit.py --f32 > $OUT/audit_f32.log 2>&1; echo "audit f32 rc=$?"; grep "behind" $OUT/audit_f32.log | cut -c1-220
timeout -k 10 900 python3 tools/dispatch_audit.py > $OUT/audit.log 2>&1; echo "audit rc=$?"; grep "behind" $OUT/audit.log | cut -c1-220
