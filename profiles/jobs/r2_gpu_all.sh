# the whole GPU suite + smoke + the pathsample driver (per-call times of the two variants)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r2a/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest_gpu.log
tail -25 gpurun_out/r2a/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2a/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r2a/smoke.log
tail -3 gpurun_out/r2a/smoke.log
( for n in 3000 40000 400000; do SPMV_HIP_STATS=1 lilac-benchmarks_amd/lib/spmvdrv pathsample lilac-benchmarks_amd/lib/libhip-spmv.so $n 100; SPMV_HIP_DISPATCH=device SPMV_HIP_STATS=1 lilac-benchmarks_amd/lib/spmvdrv pathsample lilac-benchmarks_amd/lib/libhip-spmv.so $n 100; lilac-benchmarks_amd/lib/spmvdrv pathsample oracle/_ref/native.so $n 100; done ) > gpurun_out/r2a/pathsample_driver.log 2>&1
cat gpurun_out/r2a/pathsample_driver.log
