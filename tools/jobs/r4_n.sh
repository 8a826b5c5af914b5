# round 4: is the slow x H2D of the bfs counterpart under SPMV_HIP_PIN=auto a one-time cost or a per-call one?  (16 against 64 sources)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r4n
mkdir -p $OUT
L=lilac-benchmarks_amd/lib
for pin in auto staging; do
  for src in 16 64; do
    echo "== bfs rmat:20:16, $src sources, SPMV_HIP_PIN=$pin" >> $OUT/pin2.log
    SPMV_HIP_PIN=$pin SPMV_HIP_STATS=2 timeout -k 10 200 $L/spmvdrv bfs $L/libhip-spmv.so rmat:20:16 $src >> $OUT/pin2.log 2>&1
  done
done
for pin in auto staging; do
  echo "== bfs rmat:22:16, 16 sources, SPMV_HIP_PIN=$pin" >> $OUT/pin2.log
  SPMV_HIP_PIN=$pin SPMV_HIP_STATS=2 timeout -k 10 300 $L/spmvdrv bfs $L/libhip-spmv.so rmat:22:16 16 >> $OUT/pin2.log 2>&1
  echo "== pagerank rmat:22:16, 20 iterations, SPMV_HIP_PIN=$pin" >> $OUT/pin2.log
  SPMV_HIP_PIN=$pin SPMV_HIP_STATS=2 timeout -k 10 300 $L/spmvdrv pagerank $L/libhip-spmv.so rmat:22:16 20 1 >> $OUT/pin2.log 2>&1
done
cat $OUT/pin2.log
