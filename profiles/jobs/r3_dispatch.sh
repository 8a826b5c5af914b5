cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3n
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dispatch or changing_matrices or abi_policies or cli_drivers or bench_line or pathsample or parboil or bfs" > $O/pytest_dispatch.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_dispatch.log; tail -4 $O/pytest_dispatch.log
( echo "# NPB classes S, W, A, B through the ABI in fresh processes under SPMV_HIP_DISPATCH=auto (npbcg <class> <lib> 1)"
  for c in S W A B; do SPMV_HIP_STATS=1 lilac-benchmarks_amd/lib/npbcg $c lilac-benchmarks_amd/lib/libhip-spmv.so 1 2>&1 | grep -E "Time in seconds|Mop/s total|libhip-spmv:|VERIFICATION"; echo "-- native.so"; lilac-benchmarks_amd/lib/npbcg $c oracle/_ref/native.so 1 2>&1 | grep -E "Time in seconds"; done ) > $O/npb_small_classes.log 2>&1
cat $O/npb_small_classes.log
python3 tools/ref_callers_compare.py 2>&1 | tail -6
