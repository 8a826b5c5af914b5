# round 2: host/device dispatch (f4): GPU tests, then the reference's own programs on native.so and on libhip-spmv.so
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dispatch or references or cli_drivers" > gpurun_out/r2a/pytest_dispatch.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest_dispatch.log
tail -12 gpurun_out/r2a/pytest_dispatch.log
( echo "# SPMV_HIP_DISPATCH=auto (default)"; timeout -k 10 900 python tools/ref_callers_compare.py 16 18 20
  echo "# SPMV_HIP_DISPATCH=device (round 1 behaviour)"; SPMV_HIP_DISPATCH=device timeout -k 10 900 python tools/ref_callers_compare.py 16 18 ) > gpurun_out/r2a/ref_callers.log 2>&1
cat gpurun_out/r2a/ref_callers.log
( for c in S W A B; do echo "== class $c, SPMV_HIP_DISPATCH=auto"; SPMV_HIP_STATS=1 lilac-benchmarks_amd/lib/npbcg $c lilac-benchmarks_amd/lib/libhip-spmv.so 2>&1 | tail -4; done ) > gpurun_out/r2a/npb_small_classes_auto.log 2>&1
tail -20 gpurun_out/r2a/npb_small_classes_auto.log
