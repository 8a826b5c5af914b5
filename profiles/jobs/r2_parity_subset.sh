set -x
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "striped_geometries or edge_cases or benchmark_shapes or class_c_full_size or device_resident or share_protected or kat or golden" > gpurun_out/r2a/pytest_subset.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest_subset.log
tail -15 gpurun_out/r2a/pytest_subset.log
