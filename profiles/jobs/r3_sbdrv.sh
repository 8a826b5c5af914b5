cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3sb; mkdir -p $O
( echo "# spmvdrv sb 170 (201 products through spmv_harness_, heap vectors), BLOCKED chosen automatically; SPMV_HIP_STATS=2 attributes the time"
for rep in 1 2; do SPMV_HIP_STATS=2 lilac-benchmarks_amd/lib/spmvdrv sb lilac-benchmarks_amd/lib/libhip-spmv.so 170; done
echo "# the same with SPMV_HIP_PLAN=stream (the CSR kernel, as before BLOCKED)"
SPMV_HIP_PLAN=stream SPMV_HIP_STATS=2 lilac-benchmarks_amd/lib/spmvdrv sb lilac-benchmarks_amd/lib/libhip-spmv.so 170
echo "# SPMV_HIP_STATS=0 (no synchronisation between the phases)"
for rep in 1 2; do lilac-benchmarks_amd/lib/spmvdrv sb lilac-benchmarks_amd/lib/libhip-spmv.so 170; done
echo "# the reference's native.so"
lilac-benchmarks_amd/lib/spmvdrv sb oracle/_ref/native.so 170 ) 2>&1 | tee $O/sb170_driver.log
