cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3t
timeout -k 5 200 python3 tools/striped_builder_diff.py 2>&1 | tail -14
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_builds or declines or value_edit or keeps_changing or changing_matrices or cache_and or abi_policies or no_room or diagnostics or striped_geometries or pathsample or bench_line" > gpurun_out/r3t/pytest_subset.log 2>&1; echo "rc=$?" >> gpurun_out/r3t/pytest_subset.log
tail -30 gpurun_out/r3t/pytest_subset.log
