cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/reload
SPMV_HIP_DISPATCH=device SPMV_HIP_STATS=1 SPMV_HIP_VERBOSE=1 timeout -k 10 120 lilac-benchmarks_amd/lib/spmvdrv reload lilac-benchmarks_amd/lib/libhip-spmv.so 3 > gpurun_out/reload/run.log 2>&1; echo "rc=$?" >> gpurun_out/reload/run.log
tail -30 gpurun_out/reload/run.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unloaded or timed_call or cli_drivers" 2>&1 | tail -15
