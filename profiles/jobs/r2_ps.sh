cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "churning or pathsample or keeps_changing or cache_and_write" > gpurun_out/r2a/pytest_churn.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest_churn.log
tail -12 gpurun_out/r2a/pytest_churn.log
( echo "# spmvdrv pathsample <lib> <nmin> 100: per-call times of PFOLD (the a-position array rewritten before every call) and NGT (nothing changes)"
  for n in 3000 40000 400000; do
    for mode in auto device; do echo "== nmin=$n libhip-spmv.so SPMV_HIP_DISPATCH=$mode"; SPMV_HIP_DISPATCH=$mode SPMV_HIP_STATS=1 lilac-benchmarks_amd/lib/spmvdrv pathsample lilac-benchmarks_amd/lib/libhip-spmv.so $n 100; done
    echo "== nmin=$n the reference's native.so"; lilac-benchmarks_amd/lib/spmvdrv pathsample oracle/_ref/native.so $n 100; done ) > gpurun_out/r2a/pathsample_driver.log 2>&1
cat gpurun_out/r2a/pathsample_driver.log
