cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "striped_geometries or edge_cases or benchmark_shapes or class_c_full_size or device_resident or arbitrary or non_finite" > gpurun_out/r2a/pytest_subset.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest_subset.log
tail -8 gpurun_out/r2a/pytest_subset.log
timeout -k 10 900 python tools/kbench.py --reps 50 --cases C,B,D/8,D/2 --plans striped --tunings xcd1nt0 --abl 4 > gpurun_out/r2a/kbench_pergroup.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/kbench_pergroup.log
cat gpurun_out/r2a/kbench_pergroup.log
