cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "striped_geometries or edge_cases or benchmark_shapes or class_c_full_size or arbitrary or non_finite or bfs_frontiers" > gpurun_out/r2c/pytest_quick.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2c/pytest_quick.log
tail -3 gpurun_out/r2c/pytest_quick.log
timeout -k 10 900 python tools/kbench.py --reps 100 --cases C,B,D/8 --plans striped --tunings xcd1nt0,xcd1nt0 > gpurun_out/r2c/kbench_quick.log 2>&1; grep -v "^#" gpurun_out/r2c/kbench_quick.log
