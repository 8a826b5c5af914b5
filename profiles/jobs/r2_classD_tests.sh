cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "class_d_through" -s > gpurun_out/r2a/pytest_classD.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest_classD.log
tail -12 gpurun_out/r2a/pytest_classD.log
