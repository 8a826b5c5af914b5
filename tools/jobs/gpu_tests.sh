# the whole GPU suite + smoke, as the driver runs them at round end
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r2a/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest_gpu.log
tail -25 gpurun_out/r2a/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2a/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r2a/smoke.log
tail -3 gpurun_out/r2a/smoke.log
