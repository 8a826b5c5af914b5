cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3t
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > gpurun_out/r3t/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3t/pytest_gpu.log
tail -30 gpurun_out/r3t/pytest_gpu.log
