cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
( for pol in mprotect compare; do for pin in auto staging; do echo "== SPMV_HIP_CACHE=$pol SPMV_HIP_PIN=$pin"; SPMV_HIP_PIN=$pin SPMV_HIP_CACHE=$pol SPMV_HIP_DISPATCH=device SPMV_HIP_VERBOSE=2 SPMV_HIP_STATS=2 lilac-benchmarks_amd/lib/spmvdrv pathsample lilac-benchmarks_amd/lib/libhip-spmv.so 400000 20 2>&1 | grep -v "host arrays changed" | tail -9; done; done ) > gpurun_out/r2a/ps_debug.log 2>&1
cat gpurun_out/r2a/ps_debug.log
